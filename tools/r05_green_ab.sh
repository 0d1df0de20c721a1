mkdir -p gpurun_out/r05h
python -m pytest tests -m gpu -q -x -k "schur or dual_form or golden or fp32 or every_data or direct or smoke" > gpurun_out/r05h/tests.txt 2>&1
for rep in 1 2 3; do python3 tools/ab.py "bunny_small.obj:4:64,bunny_small.obj:5:64,bunny_small.obj:3:64,bunny_small.obj:2:64" "fft=" "gemm=SHM_GREEN_GEMM=1"; done > gpurun_out/r05h/green_ab.txt 2>&1
python3 tools/ab.py "bunny_small.obj:5:32,bunny.pc:5:32,chair.obj:5:32,rocker.obj:5:32,rocker.obj:4:32,bunny_small.obj:4:32" "default=" > gpurun_out/r05h/fp32_default.txt 2>&1
