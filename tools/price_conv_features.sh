#!/bin/bash
# On the GPU box: tools/price_conv_features.sh <out file> <variant> [<variant> ...]   (variants: bits of ICN_EXP built by
# tools/build_exp.sh; "0" = the product library).  Two alternating passes over the variants inside one call.
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$1; shift
: > $OUT
for rep in 1 2; do
  for v in "$@"; do
    EXTRA_ENV=
    unset ICN_CONV_WAVES
    if [ "$v" = "0" ]; then unset ICN_LIB_PATH; tag=product;
    elif [ "$v" = "w8" ]; then unset ICN_LIB_PATH; export ICN_CONV_WAVES=8; tag=waves8;
    elif [ "$v" = "w4" ]; then unset ICN_LIB_PATH; export ICN_CONV_WAVES=4; tag=waves4;
    elif [ "${v#env:}" != "$v" ]; then unset ICN_LIB_PATH; EXTRA_ENV="${v#env:}"; tag="${v#env:}"; else export ICN_LIB_PATH=$R/geniconet_amd/csrc/build_exp/libicn_exp$v.so; tag=exp$v; fi
    env ${EXTRA_ENV:-ICN_NOOP=1} timeout -k 10 300 python3 $R/tools/price_conv_features.py --tag $tag $PRICE_ARGS >> $OUT 2>> $OUT.err || echo "variant $v failed" >> $OUT
  done
done
