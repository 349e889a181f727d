set -x
cd $GRAFT_REPO_ROOT
python bench.py --steps 20 > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err || exit 1
python bench.py --steps 10 --conv-mode mfma_f16 --no-cpu-baseline > gpurun_out/bench_f16.json 2> gpurun_out/bench_f16.err || exit 1
python bench.py --steps 10 --force-collective --no-cpu-baseline --no-other-mode > gpurun_out/bench_rccl1.json 2> gpurun_out/bench_rccl1.err || exit 1
python bench.py --steps 6 --rows 1080 --cols 1920 --batch 8 --no-cpu-baseline > gpurun_out/bench_hd8.json 2> gpurun_out/bench_hd8.err || exit 1
python bench.py --steps 4 --rows 1080 --cols 1920 --batch 16 --no-cpu-baseline --no-other-mode > gpurun_out/bench_hd16.json 2> gpurun_out/bench_hd16.err || exit 1
python bench.py --steps 50 --batch 1 --no-cpu-baseline --no-other-mode > gpurun_out/bench_b1.json 2> gpurun_out/bench_b1.err || exit 1
python bench.py --steps 200 --batch 1 --streams 4 --no-cpu-baseline --no-other-mode > gpurun_out/bench_b1s4.json 2> gpurun_out/bench_b1s4.err || exit 1
python bench.py --steps 40 --rows 1080 --cols 1920 --batch 1 --no-cpu-baseline --no-other-mode > gpurun_out/bench_hd1.json 2> gpurun_out/bench_hd1.err || exit 1
python bench.py --steps 40 --rows 1080 --cols 1920 --batch 1 --streams 3 --no-cpu-baseline --no-other-mode > gpurun_out/bench_hd1s3.json 2> gpurun_out/bench_hd1s3.err || exit 1
python bench.py --steps 8 --rows 1080 --cols 1920 --batch 8 --streams 3 --no-cpu-baseline --no-other-mode > gpurun_out/bench_hd8s3.json 2> gpurun_out/bench_hd8s3.err || exit 1
tools/profile_round.sh r03
