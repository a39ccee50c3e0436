#!/bin/bash
# usage: tools/build_variant.sh <name> <hipcc flags...>   - an experiment build of libohxgb.so (e.g. -DOHX_EXP_SSTEP=6)
# into tools/bin/variants/<name>/ (git-ignored, travels to the GPU box); tools/ab.sh runs such builds against the
# product's on the same device.
name=$1; shift
R=$(cd "$(dirname "$0")/.." && pwd)
D=$R/tools/bin/variants/$name
mkdir -p $D/obj
make -C $R/quickchem_amd/csrc -j8 LIBDIR=$D OBJDIR=$D/obj EXTRA="$*" $D/libohxgb.so 2>&1 | grep -i "error\|warning: unused\|undefined" 
ls -la $D/libohxgb.so
