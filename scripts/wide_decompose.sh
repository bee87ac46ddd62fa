#!/bin/bash
# where a k_wgemm2 tile goes: stamp builds with the LDS-DMA staging, the fragment reads, and the epilogue removed (timing-only, results wrong)
# usage: scripts/wide_decompose.sh  (libsf_{stamp,nodma,nolds,nodmalds,mfmaonly}.so built in csrc with -DSF_WEXP_STAMP [-DSF_WEXP_NODMA] [-DSF_WEXP_NOLDS] [-DSF_WEXP_NOEPI])
D=implicit-image-compression_amd/csrc
for round in 1 2; do
for v in stamp nodma nolds nodmalds mfmaonly; do
  echo "== $v"
  SIREN_FIT_LIB=$PWD/$D/libsf_$v.so timeout -k 10 120 python scripts/wide_stamp.py 2>&1 | grep -v Warning
done
done
