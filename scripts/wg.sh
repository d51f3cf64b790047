for s in 2 4 8 16 32; do echo "splits $s"; RGDA_WGRAD_SPLITS=$s python scripts/dev/dev_conv_bench.py 2>&1 | tail -14 | sed 's/ *|.*| wgrad/ wgrad/' | sed -n '6p;10p'; done
