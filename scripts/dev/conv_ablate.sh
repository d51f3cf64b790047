# which resource bounds the implicit-GEMM K loop: DMA fill only / MFMA only / MFMA + LDS reads / MFMA + fill
# (tuning build, non-pipelined 3-stage loop; results are garbage by construction, timings only)

for shape in "16 32 32 256 256 3 1 1" "16 32 32 2048 512 3 1 1"; do
  echo "pipelined (product):"; python scripts/dev/one_conv.py $shape 30 2>&1 | grep conv
  for sk in 0 2 9 1 8; do
    echo -n "NO_PIPE skip=$sk: "; RGDA_NO_PIPE=1 RGDA_CONV_SKIP=$sk python scripts/dev/one_conv.py $shape 30 2>&1 | grep conv
  done
done
