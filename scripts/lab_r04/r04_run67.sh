cd /root/repo
for i in 1 2; do
for tag in pre nb3 nb4 nb5 hip; do
  if [ $tag = hip ]; then unset DEPTHG_LIB; else export DEPTHG_LIB=$PWD/depthg_amd/lib/libdepthg_$tag.so; fi
  echo "== $tag"
  TAG=$tag scripts/kstats.sh headline 2>&1 | grep -E "k_gs|ms_per" | cut -c1-250 | sed 's/"host_ms.*//'
done; done
