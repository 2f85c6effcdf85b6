cd scripts/micro
for v in "" "-DNO_POINTS" "-DNO_DPP" "-DNO_PUBLISH" "-DNO_BARRIER" "-DNO_COMBINE" "-DNO_RETIRE" "-DNO_POINTS -DNO_DPP -DNO_PUBLISH -DNO_COMBINE -DNO_RETIRE" "-DNO_POINTS -DNO_DPP -DNO_PUBLISH -DNO_COMBINE -DNO_RETIRE -DNO_BARRIER"; do
  hipcc -O3 --offload-arch=gfx950 $v -o fps_round fps_round.hip 2>/dev/null && ./fps_round 121 "full $v"
done
