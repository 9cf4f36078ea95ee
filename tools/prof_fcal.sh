cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
rm -rf gpurun_out/fcal; mkdir -p gpurun_out/fcal
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/fcal/a -o fc -- python tools/proto/run_fetch_calib.py > gpurun_out/fcal.log 2>&1
rocprofv3 --kernel-trace --pmc TCC_MISS_sum TCC_HIT_sum --output-format csv -d gpurun_out/fcal/b -o fc -- python tools/proto/run_fetch_calib.py >> gpurun_out/fcal.log 2>&1
rocprofv3 --kernel-trace --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_BUBBLE_sum --output-format csv -d gpurun_out/fcal/c -o fc -- python tools/proto/run_fetch_calib.py >> gpurun_out/fcal.log 2>&1
python tools/proto/run_fetch_calib.py report gpurun_out/fcal
