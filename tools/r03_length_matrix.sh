#!/bin/bash
cd /root/repo; rm -rf ab
mk() { # len tag steps flags...
  local n=$1 tag=$2 steps=$3; shift 3
  KW_VARIANT_LENGTH=$n python k-wave-fluid-cuda_amd/build.py --variant L${n}_$tag "$@" > /tmp/bm2_${n}_$tag.log 2>&1 && echo "--size $n --steps $steps --warmup 3" > ab/L${n}_$tag/args || { echo "BUILD FAILED $n $tag"; rm -rf ab/L${n}_$tag; }
}
steps_of() { local n=$1; if [ $n -le 256 ]; then echo 150; elif [ $n -le 400 ]; then echo 60; elif [ $n -le 640 ]; then echo 30; else echo 10; fi; }
for n in 160 168 180 192 196 200 216 224 240 256 280 288 300 320 324 336 360 384 392 400 432 448 480 500 512 540 560 576 600 640 648 768 896 1024; do
  st=$(steps_of $n)
  for nl in 8 10 12 16; do
    mk $n b_n$nl $st -DKW_TUNE_NLX=$nl
    case $n in 196|256|324|400|576|1024) ;; *) mk $n s_n$nl $st -DKW_TUNE_SWAP -DKW_TUNE_NLX=$nl ;; esac
  done
done
mk 240 s_n12_yz12 150 -DKW_TUNE_SWAP -DKW_TUNE_NLX=12 -DKW_EXP_NLYZ=12
mk 240 s_n12_yz8 150 -DKW_TUNE_SWAP -DKW_TUNE_NLX=12 -DKW_EXP_NLYZ=8
ls ab | wc -l
