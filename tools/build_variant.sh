#!/bin/bash
# build a variant of libramx.so into ab_tmp/libramx_<tag>.so with extra compiler flags: tools/build_variant.sh T -DRAMX_CP_TIMING
set -e
tag=$1; shift
root=$(cd "$(dirname "$0")/.." && pwd)
v=/tmp/ramx_variant_$tag
rm -rf $v; mkdir -p $v/repeatafterme_amd $root/ab_tmp
cp -r $root/include $v/include
cp -r $root/repeatafterme_amd/csrc $v/repeatafterme_amd/csrc
rm -rf $v/repeatafterme_amd/csrc/build
make -s -j2 -C $v/repeatafterme_amd/csrc EXTRA="$*" ../libramx.so
cp $v/repeatafterme_amd/libramx.so $root/ab_tmp/libramx_$tag.so
echo built ab_tmp/libramx_$tag.so
