#!/bin/bash
# run scripts/gpu_updat_one.py (or $SCRIPT) for every experiment build under blocksparse_amd/variants/
SCRIPT=${SCRIPT:-scripts/gpu_updat_one.py}
for so in blocksparse_amd/variants/libbsmm_*.so; do
  BSMM_LIB=$PWD/$so python $SCRIPT "$@" 2>&1 | grep -v amdgpu.ids
done
