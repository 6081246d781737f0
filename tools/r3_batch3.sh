cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r3b3; mkdir -p $O
cp nori_amd/lib/libnori_hip.so nori_amd/lib/libnori_hip_cur.so
TIMEK=1 ROUNDS=2 bash tools/ab.sh v5 cur > $O/ab.txt 2>&1; cat $O/ab.txt
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest.txt 2>&1; tail -15 $O/pytest.txt
NORI_HIP_LIBRARY=$GRAFT_REPO_ROOT/nori_amd/lib/libnori_hip_count.so timeout 300 python tools/excursion_probe.py 8 > $O/excursions.txt 2>&1; cat $O/excursions.txt | tail -20
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 tools/ubench_valu.hip -o /tmp/ubench_valu && (cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE SQ_ACTIVE_INST_ANY SQ_THREAD_CYCLES_VALU --output-format csv -d /tmp/ub -o c -- /tmp/ubench_valu > $GRAFT_REPO_ROOT/$O/ubench.log 2>&1; find /tmp/ub -name '*counter_collection.csv' -exec cp {} $GRAFT_REPO_ROOT/$O/ubench_counter_collection.csv \; ; find /tmp/ub -name '*kernel_trace.csv' -exec cp {} $GRAFT_REPO_ROOT/$O/ubench_kernel_trace.csv \;)
python tools/pmc_summary.py $O k_ | head -120
