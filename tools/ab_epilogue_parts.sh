#!/bin/bash
# What each part of the split residual epilogue costs the launch: diagnostic builds (tools/build_variant.py epidiagN gemm_bf16.hip
# -DRR_EPI_DIAG=N; bit 0 no stores, bit 1 no residual loads, bit 2 no statistics) against the product build, alternating processes.
tag=$1
for rep in 1 2; do
  for n in 0 1 2 4 3 7; do
    lib=tools/bin/librerank_epidiag$n.so
    [ $n = 0 ] && lib=reranking-multimodal-retrievers_amd/librerank_mi355.so
    echo "== RR_EPI_DIAG=$n rep $rep"
    python tools/gemm_epilogue_timeline.py --lib $lib --no-timeline --shapes attn_out,ffn2 --rounds 4 2>/dev/null | grep product
  done
done > gpurun_out/${tag}_epilogue_parts.log
cat gpurun_out/${tag}_epilogue_parts.log
