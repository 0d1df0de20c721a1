"""Markdown rows of DESIGN.md section 6 from the bench records of a round:   python tools/r04_tables.py profiles r04"""
import json, os, sys
d, r = sys.argv[1], sys.argv[2]
def load(name):
    p = os.path.join(d, "%s_bench_%s.json" % (r, name))
    if not os.path.exists(p): return None
    t = open(p).read().strip()
    try: return json.loads(t)
    except ValueError: return json.loads(t.splitlines()[-1])
rows = [("default", "bunny_small 256^3 fp64 (configs[1])"), ("256_primal_dct", "same, primal + DCT"), ("256_primal_plain", "same, primal plain"), ("256_primal_plain_classic", "same, classic loop"),
        ("bunny_small_64_f64", "64^3"), ("bunny_small_128_f64", "128^3"), ("bunny_small_512_f64", "512^3 fp64"), ("512_primal_dct", "512 primal+DCT"), ("512_primal_plain_200", "512 primal plain 200"),
        ("512_primal_plain_200_classic", "512 classic 200"), ("bunny_pc_512_f64", "bunny.pc 512^3 fp64"), ("rocker_512_f32", "rocker 512^3 fp32"), ("rocker_512_f64", "rocker 512^3 fp64"),
        ("rocker_512_f32_primal_plain_200", "rocker fp32 primal plain 200"), ("rocker_512_f64_primal_plain_200", "rocker fp64 primal plain 200"),
        ("spraybottle_pc_1024_f32", "SprayBottle 1024 fp32"), ("spraybottle_pc_1024_f64", "SprayBottle 1024 fp64")]
for name, label in rows:
    b = load(name)
    if not b: print("| %s | (missing) |" % label); continue
    ph, s1 = b["phases_ms"], b["step1"]
    nom = s1["pairs_nominal"]
    print("| %s | %.2f ms | %.3e nodes/s | conv %.2f | wait %.2f | pcg %.2f (%d its, %.4f ms/it, loop frac %.3f, project %.3f) | step1 frac %.3f pairs f64 %.3f f32 %.3f | kernels %s |" % (
        label, b["ms_per_step"], b["value"], ph["ms_conv"], ph["ms_wait_setup"], ph["ms_pcg"], b["config"]["cg_iters"], b["pcg"]["ms_per_iter"], b["pcg"]["frac_of_hbm_peak"], b["pcg"]["ms_project_avg"],
        s1["frac"] or 0, s1["pairs_fp64"] / nom, s1["pairs_fp32"] / nom, {k: round(v["frac_of_hbm_peak"] or 0, 3) for k, v in b["kernels"].items()}))
b = load("default")
if b and "also" in b:
    for k, v in b["also"].items():
        if k.startswith("stencil"):
            print("also.%s: m %d, %s, ms/iter %.4f, loop %.3f, project %.3f, %s" % (k, v["constraint_rows"], v["projector"], v["ms_per_iter"], v["loop_frac_of_hbm_peak"], v["ms_project_avg"], {kk: round(vv["frac_of_hbm_peak"], 3) for kk, vv in v["kernels"].items()}))
        else:
            print("also.%s: %s" % (k, {kk: vv for kk, vv in v.items() if kk in ("value", "ms_per_step", "phases_ms", "linf_fp32_vs_fp64", "cg_iters")}))
    print("cpu_baseline", {k: v for k, v in b["cpu_baseline"].items() if k != "sample"})
    print("roofline", {k: v for k, v in b["roofline"].items() if k in ("kernel", "achieved", "peak", "frac", "traffic")})
