cd $GRAFT_REPO_ROOT
tools/prof_channels.sh r05_chan_plain 1 --placement plain > /dev/null 2>&1
tools/prof_channels.sh r05_chan_pair 1 > /dev/null 2>&1
for t in plain pair; do for g in 1 2; do echo "== $t group $g"; grep -A4 "fdg_isa_eval_nt" gpurun_out/prof_r05_chan_$t/channels_proc1_grp$g.txt | tail -15; done; done
