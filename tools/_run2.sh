set -x
python -m pytest tests/test_gpu_tiling.py -x -q > gpurun_out/r2_pytest_tiles.log 2>&1; echo exit=$? >> gpurun_out/r2_pytest_tiles.log; tail -5 gpurun_out/r2_pytest_tiles.log
hipcc --offload-arch=gfx950 -O3 -w -o /tmp/valu_rate tools/ubench/valu_rate.hip && timeout -k 5 120 /tmp/valu_rate > gpurun_out/r2_valu_rate.txt 2>&1; cat gpurun_out/r2_valu_rate.txt
timeout -k 10 300 python bench.py --mode tiles --steps 24 --warmup 4 > gpurun_out/r2_tiles1.json 2> gpurun_out/r2_tiles1.err; echo tiles1=$?; cat gpurun_out/r2_tiles1.json; tail -3 gpurun_out/r2_tiles1.err
SGM_BENCH_BACKEND=gloo timeout -k 10 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 3 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 3 --mode tiles --steps 24 --warmup 4 > gpurun_out/r2_tiles3.json 2> gpurun_out/r2_tiles3.err; echo tiles3=$?; cat gpurun_out/r2_tiles3.json; tail -5 gpurun_out/r2_tiles3.err
timeout -k 10 300 python bench.py --workload uhd_3840x2160_d128_p8 --batch 1 --steps 24 --warmup 4 --no-cpu-baseline --no-host-boundary > gpurun_out/r2_uhd_frames.json 2> gpurun_out/r2_uhd_frames.err; echo uhd=$?; cat gpurun_out/r2_uhd_frames.json
