cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:?}"
timeout -k 10 600 python -m pytest tests/test_conv3_rw_gpu.py -q -x 2>&1 | tail -3 || exit 1
for pc in g_bwd; do
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/pt_$pc -o t -- python3 tools/chain_trace.py run $pc > gpurun_out/pt_$pc.log 2>&1
f=$(find gpurun_out/pt_$pc -name "*kernel_trace.csv" | head -1)
python3 tools/chain_trace.py parse $f 20 > gpurun_out/piece_${pc}_new.log 2>&1
rm -rf gpurun_out/pt_$pc
done
bash tools/ab_libs.sh build_tmp/lib_old.so
