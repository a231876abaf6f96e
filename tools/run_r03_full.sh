#!/bin/bash
# round 3 on the GPU box: suite + smoke + bench line + rocprofv3 evidence + call-path configurations + CLI rates (TAG under gpurun_out/)
set -u
export TAG=${TAG:-r03f}
O=$PWD/gpurun_out/$TAG; mkdir -p $O
bash tools/run_r03.sh 2>&1 | tail -14
bash tools/profile.sh ${TAG}_prof > $O/profile.log 2>&1; tail -6 $O/profile.log
bash tools/traffic.sh ${TAG}_traffic ped10 2>&1 | tail -2
bash tools/call_cfg_r03.sh > $O/call_cfg.txt 2>&1; cat $O/call_cfg.txt
cd $GRAFT_REPO_ROOT
python tools/cli_throughput.py 3000000 > $O/cli_3M.txt 2>&1; grep "sites/s\|loop" $O/cli_3M.txt | head -12
du -sh gpurun_out
