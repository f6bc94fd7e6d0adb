set -e -o pipefail
cd $GRAFT_REPO_ROOT
O=gpurun_out
for e in "RIB_SPADE_DMA=0" "RIB_SPADE_DMA=2" "RIB_SPADE_DMA=2 RIB_SPADE_DMA_NF=2" "RIB_SPADE_DMA=2 RIB_SPADE_DMA_NF=4" "RIB_SPADE_DMA=2 RIB_SPADE_DMA_WGS=512" "RIB_SPADE_DMA=2 RIB_SPADE_DMA_WGS=1024" "RIB_SPADE_DMA=2 RIB_SPADE_DMA_WGS=2048"; do echo "## $e"; env $e python tools/time_ops.py .spade 2>/dev/null | grep -v modulate | cut -c1-60,108-130; done | tee $O/r04_spade_dma_ops.txt
