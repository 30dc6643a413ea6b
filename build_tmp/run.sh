cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:?}"
timeout -k 10 600 python -m pytest tests/test_kernels_gpu.py -q -x -k "content or loss" 2>&1 | tail -3 || exit 1
for lib in "$PWD/build_tmp/lib_old.so" ""; do
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/pt_tail -o t -- python3 tools/chain_trace.py run chain_tail > gpurun_out/pt_tail.log 2>&1
f=$(find gpurun_out/pt_tail -name "*kernel_trace.csv" | head -1)
TECOGAN_LIB=$lib python3 tools/chain_trace.py parse $f 20 2>&1 | tail -4
rm -rf gpurun_out/pt_tail
done
bash tools/ab_libs.sh build_tmp/lib_old.so 2>&1 | grep -E "== lib|whole step|bench"
