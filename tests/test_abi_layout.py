"""The ctypes mirrors of the C ABI structs against the headers themselves: size of every struct and offset of every field, taken from
a C program compiled against include/*.h (and include/kq_cycle_tas.h for the cycle's TAS boundary). A field added to a header and not
to its mirror makes the library read past the end of the Python-built struct — this is the check for that."""
import ctypes as C
import os
import re
import subprocess
import tempfile

from kueue_amd import _ffi as F
from kueue_amd import tas as T
from kueue_amd import tas_cycle as TC

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
STRUCTS = {
    "kq_config": F.kq_config, "kq_snapshot": F.kq_snapshot, "kq_heads": F.kq_heads, "kq_pending": F.kq_pending, "kq_decisions": F.kq_decisions,
    "kq_afs_ledger": F.kq_afs_ledger, "kq_row_patch": F.kq_row_patch,
    "kq_tas_topology": T.kq_tas_topology, "kq_tas_requests": T.kq_tas_requests, "kq_tas_result": T.kq_tas_result, "kq_tas_replacement": T.kq_tas_replacement,
    "kq_cycle_tas": TC.kq_cycle_tas, "kq_cycle_tas_out": TC.kq_cycle_tas_out,
}


def test_ctypes_mirrors_match_the_headers():
    lines = ['#include <stdio.h>', '#include <stddef.h>', '#include "kq_engine.h"', '#include "kq_tas.h"', '#include "kq_cycle_tas.h"',
             'int main(void) {']
    for name, cls in STRUCTS.items():
        lines.append(f'  printf("{name} %zu\\n", sizeof({name}));')
        for fname, _ in cls._fields_:
            lines.append(f'  printf("{name}.{fname} %zu\\n", offsetof({name}, {fname}));')
    lines += ['  return 0;', '}']
    with tempfile.TemporaryDirectory() as d:
        src = os.path.join(d, "layout.c")
        with open(src, "w") as fh:
            fh.write("\n".join(lines))
        exe = os.path.join(d, "layout")
        subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), "-I", os.path.join(ROOT, "oracle"), "-o", exe, src])
        out = subprocess.check_output([exe], text=True)
    got = dict(l.split() for l in out.strip().splitlines())
    for name, cls in STRUCTS.items():
        assert int(got[name]) == C.sizeof(cls), (name, got[name], C.sizeof(cls))
        for fname, _ in cls._fields_:
            assert int(got[f"{name}.{fname}"]) == getattr(cls, fname).offset, (name, fname)


def test_every_header_field_has_a_mirror_field():
    """Field COUNT per struct, read from the header text (a mirror that merely stops early would pass the offset check)."""
    text = ""
    for h in ("include/kq_engine.h", "include/kq_tas.h", "include/kq_cycle_tas.h"):
        text += open(os.path.join(ROOT, h)).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    for name, cls in STRUCTS.items():
        m = re.search(r"typedef struct " + name + r"\s*\{(.*?)\}\s*" + name + r"\s*;", text, flags=re.S)
        assert m, name
        n = 0
        for decl in m.group(1).split(";"):
            decl = decl.strip()
            if decl:
                n += decl.count(",") + 1
        assert n == len(cls._fields_), (name, n, len(cls._fields_))
