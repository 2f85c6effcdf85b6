cd /root/repo
for tag in hip hs640 hs768 hs896; do
  if [ $tag = hip ]; then unset DEPTHG_LIB; else export DEPTHG_LIB=$PWD/depthg_amd/lib/libdepthg_$tag.so; fi
  echo "== $tag"; TAG=$tag bash scripts/kstats.sh headline+head 2>&1 | grep -E "wgrad|reduce|ms_per_step" | cut -c1-110
done
