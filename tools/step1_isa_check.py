#!/usr/bin/env python3
"""Instruction counts of the two hot loops of the Step-1 kernel, from the compiler's own assembly (no GPU): the near loop (one fp64 source against a lane's four nodes:
four v_rsq_f64) and the far loop (four packed-fp32 sources against them: sixteen v_exp_f32), for the fp64 and the fp32 solve's instantiation.
Round 6 found the far loop of a build carrying 24 s_nop of trans-use hazards (142 instructions where 116 were possible: +8 % on the fp32 solve) after UNRELATED code around it
had changed -- tests/test_abi_and_host.py holds the loops to their measured-good shapes the way it holds the registers.
    python tools/step1_isa_check.py            # prints one line per loop, JSON on the last line"""
import collections
import json
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "signed-heat-3d_amd", "csrc")
SRC = """#include <hip/hip_runtime.h>
#include "shm_conv_tiered.hip.h"
namespace shm {
template __global__ void conv_tiered_kernel<4, double, true>(ConvParams, const double*, const float*, const double*, double*, double*, double*, unsigned long long*, unsigned*);
template __global__ void conv_tiered_kernel<4, float, false>(ConvParams, const double*, const float*, const double*, float*, float*, float*, unsigned long long*, unsigned*);
}
"""


def loops():
    with tempfile.TemporaryDirectory() as t:
        src, asm = os.path.join(t, "k.hip"), os.path.join(t, "k.s")
        open(src, "w").write(SRC)
        subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "--cuda-device-only", "-S", "-I" + CSRC, "-I" + os.path.join(ROOT, "include"), src, "-o", asm],
                              stderr=subprocess.DEVNULL)
        lines = open(asm).read().split("\n")
    out = {}
    for key, name in (("ILi4EdLb1", "fp64 solve"), ("ILi4EfLb0", "fp32 solve")):
        start = next(i for i, l in enumerate(lines) if l.startswith("_ZN3shm18conv_tiered_kernel" + key))
        end = next(i for i in range(start, len(lines)) if "s_endpgm" in lines[i])
        cur, blocks = None, {}
        for i in range(start, end):
            m = re.match(r"^(\.LBB\d+_\d+):", lines[i])
            if m:
                cur = m.group(1)
                blocks[cur] = []
            elif cur and lines[i].strip() and not lines[i].strip().startswith(";") and not lines[i].strip().startswith("."):
                blocks[cur].append(lines[i].split()[0])
        for ops in blocks.values():
            c = collections.Counter(ops)
            if c["v_rsq_f64_e32"] == 4:
                out[name + " near loop"] = {"instructions": len(ops), "s_nop": c["s_nop"], "valu": sum(v for k, v in c.items() if k.startswith("v_"))}
            if c["v_exp_f32_e32"] == 16:
                out[name + " far loop"] = {"instructions": len(ops), "s_nop": c["s_nop"], "valu": sum(v for k, v in c.items() if k.startswith("v_"))}
    return out


if __name__ == "__main__":
    r = loops()
    for k, v in r.items():
        print("%-22s %3d instructions, %2d s_nop, %3d vector" % (k, v["instructions"], v["s_nop"], v["valu"]))
    print(json.dumps(r))
    sys.exit(0 if len(r) == 4 else 1)
