#!/bin/bash
# usage: tools/build_variant.sh <name> <hipcc flags...>   - an experiment build of libohxgb.so (e.g. -DOHX_EXP_SSTEP=6)
# into tools/bin/variants/<name>/ (git-ignored, travels to the GPU box); tools/ab.sh runs such builds against the
# product's on the same device.
# An experiment that changes SOURCE rather than a -D: copy quickchem_amd/csrc to quickchem_amd/.exp_<name>/ (git-ignored; the
# same depth, so the Makefile's ../../include resolves), edit there, and build it with
#   make -C quickchem_amd/.exp_<name> LIBDIR=$PWD/tools/bin/variants/<name> OBJDIR=$PWD/tools/bin/variants/<name>/obj $PWD/tools/bin/variants/<name>/libohxgb.so
# - the shipped sources, and with them the hash profiles/*_traffic.json is tagged with, stay as they are until something
# is kept; diff -u against csrc/ is the patch for profiles/ (round 6: r06_not_kept_idxen_gather.patch was made this way).
name=$1; shift
R=$(cd "$(dirname "$0")/.." && pwd)
D=$R/tools/bin/variants/$name
mkdir -p $D/obj
make -C $R/quickchem_amd/csrc -j8 LIBDIR=$D OBJDIR=$D/obj EXTRA="$*" $D/libohxgb.so 2>&1 | grep -i "error\|warning: unused\|undefined" 
ls -la $D/libohxgb.so
