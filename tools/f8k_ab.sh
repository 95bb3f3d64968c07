B="python3 tools/bench_conv.py --only 0:128:64 --base 64 --levels 5 --f8 1 --wgrad 0 --iters 30"
L=interactive-unet_amd/lib
run() { echo "== $1"; shift; env "$@" timeout -k 10 120 $B 2>&1 | grep "fp8"; }
run main512 A=1
run nlt256 IUNET_LIB=$L/libiunet_ab256.so
run nomfma IUNET_LIB=$L/libiunet_abnomfma.so
run noconv IUNET_F8K_DBG=1
run noactload IUNET_F8K_DBG=2
run noweights IUNET_F8K_DBG=8
run noact_noweights IUNET_F8K_DBG=10
run nomfma_noact_nowe IUNET_F8K_DBG=10 IUNET_LIB=$L/libiunet_abnomfma.so
