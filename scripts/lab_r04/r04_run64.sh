cd /root/repo
export DEPTHG_LIB=$PWD/depthg_amd/lib/libdepthg_gsdev.so
for dbg in 0 64 32 96 2048 2144; do
  echo "== DG_DEBUG=$dbg (64: every G tile read is tile 0 of the row; 32: no transposing reads / MFMAs; 2048: no epilogue)"
  DG_DEBUG=$dbg TAG=gs$dbg bash scripts/kstats.sh headline 2>&1 | grep -E "k_gs|k_corr2" | cut -c1-110
done
