cd /root/repo
for hw in 56 28 40; do
python scripts/r04_c5dbg.py $hw 2>&1 | tail -1
DEPTHG_LIB=$PWD/depthg_amd/lib/libdepthg_oldred.so python scripts/r04_c5dbg.py $hw 2>&1 | tail -1
done
