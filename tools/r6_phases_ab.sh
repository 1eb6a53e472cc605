cd $GRAFT_REPO_ROOT; O=gpurun_out/r6h; mkdir -p $O
for cfg in "MRFA_PROLOGUE_FUSION=1" "MRFA_PROLOGUE_FUSION=0" "MRFA_PROLOGUE_FUSION=0 MRFA_WGRAD_LEAN=0"; do
  tag=$(echo $cfg | tr ' =' '__'); env $cfg python tools/step_phases.py 8 mtia 20 2>/dev/null | grep -v amdgpu > $O/phases_$tag.txt; echo "== $cfg"; grep -E "replayed|keypoints|generated|TokenPose_B|wgrads|stage3.3|stage2|optimizer: done" $O/phases_$tag.txt
done
