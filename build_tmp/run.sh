cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:?}"
timeout -k 10 900 python -m pytest tests/test_kernels_gpu.py -q -x -k "conv4s2 or convt or s2 or random_shapes or dgrad" 2>&1 | tail -3 || exit 1
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/pt -o t -- python3 tools/chain_trace.py run g_bwd > gpurun_out/pt.log 2>&1
f=$(find gpurun_out/pt -name "*kernel_trace.csv" | head -1)
python3 tools/chain_trace.py parse $f 20 2>&1 | sed -n 1,18p
rm -rf gpurun_out/pt
bash tools/ab_libs.sh build_tmp/lib_old.so 2>&1 | grep -E "== lib|g_bwd alone|whole step|bench"
