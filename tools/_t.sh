export TMPDIR=/tmp
mkdir -p gpurun_out/sb
timeout 900 python -m pytest tests/test_gpu_sift.py tests/test_gpu_sift_matcher.py -x -q 2>&1 < /dev/null | tail -3
timeout 900 python -m pytest tests/test_gpu_fuzz.py -x -q -k sift 2>&1 < /dev/null | tail -2
rm -rf /tmp/pp; timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/pp -o t -- python bench.py --workload cfg2 --steps 3 --warmup 1 --no-cpu-baseline > /tmp/pp.log 2>&1 < /dev/null
timeout 60 python profiles/summarize_rocpd.py /tmp/pp/t_results.db < /dev/null | grep -E "sift_base|sift_half|gray_kernel|total kernel" | cut -c1-150
timeout 400 python bench.py --workload cfg2 --steps 6 --warmup 2 --no-cpu-baseline > gpurun_out/sb/cfg2.json 2>&1 < /dev/null
