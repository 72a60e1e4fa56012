#!/bin/bash
# Build-time A/B: a second copy of the tree under _alt/<name>/ (git-ignored, travels to the GPU box) built with extra
# compiler flags (the copy leaves out etol_amd/lib and every .so: with the main build's objects copied along, timestamps
# intact, make found nothing to rebuild and the 'variant' was the main build -- round-2 A/Bs made with this script before that
# was fixed compared a build with itself), so that two builds can be timed on the SAME box in one gpurun call (boxes differ by up to 10 %).
#   bash tools/ab_build.sh waves5 -DEMI_PASS_WAVES_PER_EU=5
#   gpurun -- 'python tools/pass_variants.py ... ; (cd _alt/waves5 && python tools/pass_variants.py ...)'
set -e
name=$1; shift
R=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p $R/_alt/$name
rm -rf $R/_alt/$name && mkdir -p $R/_alt/$name
(cd $R && tar cf - --exclude=./_alt --exclude=./.git --exclude=./gpurun_out --exclude=./oracle/_ref --exclude=__pycache__ --exclude=./profiles --exclude=./tests --exclude=./etol_amd/lib --exclude='*.so' .) | tar xf - -C $R/_alt/$name
make -C $R/_alt/$name -j6 lib EXTRA_HIPFLAGS="$*" > $R/_alt/$name/build.log 2>&1 || { tail $R/_alt/$name/build.log; exit 1; }
echo "built _alt/$name with: $*"
