mkdir -p gpurun_out
(timeout 600 python tools/bench_encoder.py 2>&1 | grep -v amdgpu.ids | tail -12) > gpurun_out/bench_encoder.txt; cat gpurun_out/bench_encoder.txt
