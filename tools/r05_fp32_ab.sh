mkdir -p gpurun_out/r05g
for rep in 1 2; do python3 tools/ab.py "SprayBottle.pc:4:32,rocker.obj:5:32,bunny_small.obj:4:32,bunny.pc:5:32,SprayBottle.pc:6:32" "tiered32=" "classic=SHM_CONV32_CLASSIC=1"; done > gpurun_out/r05g/ab.txt 2>&1
python -m pytest tests -m gpu -x -q -k "32 or f32 or fp32 or precision or every_data or culled or slab or stress" > gpurun_out/r05g/tests32.txt 2>&1
