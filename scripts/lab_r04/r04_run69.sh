cd /root/repo
for i in 1 2; do
for hack in 0 1; do
  if [ $hack = 1 ]; then export DG_HACK_BF16F=1; else unset DG_HACK_BF16F; fi
  echo "== hack $hack"
  TAG=h$hack scripts/kstats.sh headline+head 2>&1 | grep -E "k_head|ms_per" | cut -c1-250 | sed 's/"host_ms.*//'
done; done
