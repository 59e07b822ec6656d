// kq_tas_cycle_kernel_bal.hip — k_nominate_tas_bal / k_process_tas_bal: the kernels of kq_cycle_run_tas once more, with tas_balanced_placement.go
// inside the placement (kq_tas_device.hpp t_balanced_lane0). Launched instead of the plain ones when a TAS flavor of the cycle carries
// KQ_TAS_F_BALANCED_PLACEMENT (features.TASBalancedPlacement, default off), so that the plain kernels do not pay for the gate.
#define KQ_TAS_BAL 1
#include "kq_tas_cycle_kernel.hip"
