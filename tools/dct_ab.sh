#!/bin/bash
# A/B of the transform-kernel variants built by tools/dct_variants.sh (GPU box): dual-solver ms per iteration at 256^3 and 512^3
cd $GRAFT_REPO_ROOT
for v in v0 hint lc4 lc4_hint hoist_hint twg_hint lc4_twg_hint; do
  for w in bunny_small_256_f64 bunny_small_512_f64; do
    SHM_GRID_LIB=$GRAFT_REPO_ROOT/signed-heat-3d_amd/lib/variants/libshm_grid_$v.so python bench.py --workload $w --no-cpu-baseline --steps 4 --warmup 2 > gpurun_out/ab_${v}_$w.json 2> gpurun_out/ab_${v}_$w.err
  done
done
python - <<PY
import json
for v in "v0 hint lc4 lc4_hint hoist_hint twg_hint lc4_twg_hint".split():
    row=[v]
    for w in ("bunny_small_256_f64","bunny_small_512_f64"):
        try:
            d=json.loads(open("gpurun_out/ab_%s_%s.json"%(v,w)).read().strip().splitlines()[-1])
            row.append("%s: %.2f ms, pcg %.2f, %.4f ms/it (%d)"%(w[12:15], d["ms_per_step"], d["phases_ms"]["ms_pcg"], d["pcg"]["ms_per_iter"], d["config"]["cg_iters"]))
        except Exception as e:
            row.append("FAIL")
    print(" | ".join(row))
PY
