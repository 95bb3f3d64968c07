# A/B and ablation runs of the K = 128 fp8 conv on one layer (e4m3 planes in and out).  bash tools/f8k_ab.sh [L:Cin:Cout]
# needs (every A/B library built with -DIUNET_ABLATE, which compiles the IUNET_F8K_DBG switches in): tools/ab_build.sh conv3_f8k.hip -DIUNET_ABLATE -> lib/libiunet_ab.so (IUNET_LIB), -DF8K_NLT8=512 -> lib/libiunet_ab512.so, -DF8K_NOMFMA -> lib/libiunet_abnomfma.so
LAYER=${1:-0:128:64}
B="python3 tools/bench_conv.py --only $LAYER --base 64 --levels 5 --f8 2 --wgrad 0 --iters 30"
L=interactive-unet_amd/lib
run() { echo "== $1"; shift; env "$@" timeout -k 10 120 $B 2>&1 | grep "fp8"; }
run main A=1
run loaders512 IUNET_LIB=$L/libiunet_ab512.so
run nomfma IUNET_LIB=$L/libiunet_abnomfma.so
run noactload IUNET_F8K_DBG=2
run noweights IUNET_F8K_DBG=8
run noact_noweights IUNET_F8K_DBG=10
run nomfma_noact_nowe IUNET_F8K_DBG=10 IUNET_LIB=$L/libiunet_abnomfma.so
