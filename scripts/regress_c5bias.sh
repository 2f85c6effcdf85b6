#!/bin/bash
# Can the committed -m gpu suite see round 4's dropped-MFMA bug?  Builds depthg_amd/lib/libdepthg_c5rev.so = the production library with
# scripts/experiments/k_corr2_c5bias_revert.patch applied to dg_corr2.hip (the pre-fix statement), here in the build container:
#     scripts/regress_c5bias.sh build
# and, on the GPU box, runs the tests that bound the loss means on grids without padded positions against it (they must FAIL) and
# against the production library (they must PASS):
#     scripts/regress_c5bias.sh run > profiles/r05_c5bias_regression.txt
set -e
root="$(cd "$(dirname "$0")/.." && pwd)"
csrc=$root/depthg_amd/csrc
TESTS="tests/test_gpu_configs.py::test_config5_hires_56_vs_oracle tests/test_gpu_boundary.py::test_exact_masks_on_the_dense_grid tests/test_gpu_sweep.py::test_dense_grids_without_padded_positions"
case "$1" in
build)
    make -C $csrc -j4 >/dev/null
    work=$root/depthg_amd/lib/obj_c5rev
    mkdir -p $work
    cp $csrc/dg_corr2.hip $work/dg_corr2.hip
    patch -s -p3 -d $work -i $root/scripts/experiments/k_corr2_c5bias_revert.patch
    hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -mllvm -disable-machine-licm -I$csrc -I$root/include \
        -c $work/dg_corr2.hip -o $work/dg_corr2.o
    objs=$(ls $root/depthg_amd/lib/obj/*.o | grep -v "/dg_corr2.o")
    hipcc -shared -fPIC --offload-arch=gfx950 $objs $work/dg_corr2.o -o $root/depthg_amd/lib/libdepthg_c5rev.so
    echo built depthg_amd/lib/libdepthg_c5rev.so
    ;;
run)
    cd $root
    echo "== production library: the tests must pass"
    python -m pytest $TESTS -q 2>&1 | tail -4
    echo "== with k_corr2_c5bias_revert.patch (the round-4 bug back in): the tests must fail"
    DEPTHG_LIB=$root/depthg_amd/lib/libdepthg_c5rev.so python -m pytest $TESTS -q 2>&1 | grep -E "^(FAILED|ERROR)|passed|failed" | cut -c1-260
    ;;
*) echo "usage: $0 build|run"; exit 2;;
esac
