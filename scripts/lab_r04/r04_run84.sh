cd /root/repo
for s in 277 290 386; do
  echo "== seed $s HEAD"; python scripts/fuzz_parity.py 1 $s 2>&1 | grep -E "FAIL|cases in" | cut -c1-160
  echo "== seed $s before the gather merges"; DEPTHG_LIB=$PWD/depthg_amd/lib/libdepthg_old.so python scripts/fuzz_parity.py 1 $s 2>&1 | grep -E "FAIL|cases in" | cut -c1-160
done
