#!/bin/bash
# developer aid: k_head_wgrad3's ablation builds next to the product build, one box (WRONG results in the ablations; kernel-trace averages)
#   here:  for v in "b:-DW3_ABL_NOSTORE -DW3_ABL_NODMA" ...; do make -C depthg_amd/csrc EXTRA=... OBJDIR=../lib/obj_w3$k LIB=../lib/libdepthg_w3$k.so; done
#   box:   scripts/lab_r06/w3_ablate.sh
for v in base b c a; do
  lib=/root/repo/depthg_amd/lib/libdepthg_w3$v.so
  [ $v = base ] && lib=/root/repo/depthg_amd/lib/libdepthg_hip.so
  echo "== $v"
  /root/repo/scripts/lab_r06/head_table.sh head_w3$v $lib | grep -E "step|wgrad3|head_dh"
done
