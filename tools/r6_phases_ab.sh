#!/bin/bash
# step timeline (tools/step_phases.py) under several environment settings on one box:  PHCFGS="A=1 B=0" bash tools/r6_phases_ab.sh <tag>
cd $GRAFT_REPO_ROOT; O=gpurun_out/$1; mkdir -p $O
for cfg in ${PHCFGS}; do
  tag=$(echo $cfg | tr ' =' '__'); env $cfg python tools/step_phases.py 8 mtia 20 2>/dev/null | grep -v amdgpu > $O/phases_$tag.txt; echo "== $cfg"; grep -E "replayed|keypoints|generated|TokenPose_B|wgrads|stage3.3|stage2|optimizer: done" $O/phases_$tag.txt
done
