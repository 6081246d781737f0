# round 5, GPU call 19: with the tails beside the next batch, from how many live paths should a batch be handed to wf_finish? (pa5 table, 4 batches)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r5_19; mkdir -p $O
WORKLOAD=c4-table-mis SPP=512 timeout 500 python tools/tail_probe.py > $O/finish_paths_with_overlap_c4.txt 2>&1; cat $O/finish_paths_with_overlap_c4.txt
echo "t = $SECONDS s"
