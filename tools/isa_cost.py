"""Weighted instruction histogram of a region of an AMDGPU assembly listing (hipcc -S): python tools/isa_cost.py FILE.s FIRST LAST
Costs = issue cycles of a wave64 instruction on a gfx950 SIMD with two waves resident, measured with tools/coexec_probe.hip
(vector instructions and MFMAs of a SIMD do not overlap, so a loop body's cycles add up)."""
import collections
import re
import sys

HALF = ("v_max_f32", "v_min_f32", "v_cvt_pk", "v_mul_lo", "v_mul_hi", "v_bfe", "v_med3", "v_add3", "v_lshl_add_u32", "v_bitop3", "v_perm", "v_and_or",
        "v_or3", "v_lshl_or", "v_mad_u32", "v_mad_i32", "v_mad_u64", "v_mad_i64", "v_fma_f32", "v_bfi", "v_alignbit", "v_cndmask_b32_e64", "v_cmp_class")


def cost(op):
    if op.startswith("v_mfma"):
        return 16.5 if "16x16x32" in op else (8.5 if "16x16x16" in op else 32.5)
    if op.startswith("v_pk_"):
        return 4.7
    if "u64" in op or "i64" in op or "b64" in op:
        return 6.5
    if op.startswith("v_") and any(op.startswith(h) for h in HALF):
        return 4.4
    if op.startswith("v_"):
        return 2.5
    return 0.0


lines = open(sys.argv[1]).read().split("\n")[int(sys.argv[2]) - 1:int(sys.argv[3])]
ops = collections.Counter()
for ln in lines:
    m = re.match(r"\s+([a-z_0-9]+)", ln)
    if m and not ln.strip().startswith(";"):
        ops[m.group(1)] += 1
tot = collections.Counter()
for op, n in ops.items():
    kind = "mfma" if op.startswith("v_mfma") else ("valu" if op.startswith("v_") else ("lds" if op.startswith("ds_") else ("vmem" if op.startswith(("global_", "scratch_", "buffer_")) else "salu/other")))
    tot[kind] += n * cost(op) if kind in ("mfma", "valu") else n
print({k: round(v, 1) for k, v in tot.items()})
for op, n in sorted(ops.items(), key=lambda kv: -kv[1] * max(cost(kv[0]), 0.01))[:28]:
    print(f"{n:5d} x {cost(op):4.1f} = {n * cost(op):7.1f}  {op}")
