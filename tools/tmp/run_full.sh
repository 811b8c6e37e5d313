python -m pytest tests -m gpu -x -q > gpurun_out/r04_gputests_b.log 2>&1; tail -5 gpurun_out/r04_gputests_b.log
(time python bench.py --steps 20 --warmup 5) > gpurun_out/r04_bench_b.json 2> gpurun_out/r04_bench_b.err; tail -4 gpurun_out/r04_bench_b.err
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && bash tools/profile.sh r04b > gpurun_out/r04_profile_b.log 2>&1; tail -5 gpurun_out/r04_profile_b.log
