cd /root/repo
for lib in gsdevpre gsdev; do
export DEPTHG_LIB=$PWD/depthg_amd/lib/libdepthg_$lib.so
for dbg in 0 8192 16384 2144 10336; do
  echo "== $lib DG_DEBUG=$dbg (8192: no depth blocks; 16384: depth blocks alone; 2144: stream blocks without G bytes / MFMAs / epilogue; 10336 = 2144 + 8192)"
  DG_DEBUG=$dbg TAG=gs$dbg bash scripts/kstats.sh headline 2>&1 | grep -E "k_gs|k_corr2" | cut -c1-110
done; done
