cd /root/repo; mkdir -p gpurun_out/r04
DG_HEAD_STAMPS=$PWD/gpurun_out/r04/head_stamps.txt DEPTHG_LIB=$PWD/depthg_amd/lib/libdepthg_hstamp.so python scripts/head_phase.py 2>&1 | tail -20
