#!/usr/bin/env bash
# run_quant.sh -- launcher with the reference's env-var surface (quant/gptq/run_quant.sh:1-40).
# One process per GPU over RCCL/xGMI.  HIP_VISIBLE_DEVICES (or CUDA_VISIBLE_DEVICES) picks the GPUs.
set -euo pipefail

BITS=${1:-Q4_K}
export OMP_NUM_THREADS=8
export GPU_MAX_HW_QUEUES="${GPU_MAX_HW_QUEUES:-16}"  # one hardware queue per HIP stream
export HSA_ENABLE_IPC_MODE_LEGACY=0

DEVICES="${HIP_VISIBLE_DEVICES:-${CUDA_VISIBLE_DEVICES:-0}}"
NUM_GPUS=$(echo "$DEVICES" | tr ',' '\n' | wc -l)
MASTER_PORT="${MASTER_PORT:-29500}"
HERE="$(cd "$(dirname "${BASH_SOURCE[0]}")" && pwd)"

python -m torch.distributed.run --nnodes=1 --nproc-per-node="$NUM_GPUS" --master-addr 127.0.0.1 \
    --master-port "$MASTER_PORT" "$HERE/quant.py" \
    --model_name_or_path "${MODEL:-meta-llama/Llama-3.2-1B-Instruct}" \
    ${TOKENIZER_NAME:+--tokenizer_name "$TOKENIZER_NAME"} \
    --quantizable_modules '.*layers.*((q|k|v|o|gate|up|down)_proj)$' \
    --pre_block_modules model.embed_tokens \
    --block_modules model.layers \
    --post_block_modules lm_head \
    --quant_non_block_modules \
    --calibration_data "${CALIB_DATA:?set CALIB_DATA to a .pt file of [1, L] token-id tensors (dataset downloads are not part of this package)}" \
    --calibration_tokens "${CALIB_TOKENS:-4194304}" \
    --calibration_sequence_length "${CALIB_SEQ_LEN:-4096}" \
    --quant_scale "${QUANT_SCALE:-absmax}" \
    --rel_damp "${REL_DAMP:-0.01}" \
    --block_size "${BLOCK_SIZE:-128}" \
    --default_bit_width "${BITS:-Q4_K}" \
    ${BIT_WIDTH_CONFIGURATION:+--bit_width_configuration "$BIT_WIDTH_CONFIGURATION"} \
    --rmin "${RMIN:--1.0}" \
    --rdelta "${RDELTA:-0.1}" \
    --nstep "${NSTEP:-20}" \
    --dtype "${DTYPE:-auto}" \
    --seed "${SEED:-0}" \
    ${ATTN_IMPL:+--attn_implementation "$ATTN_IMPL"} \
    --verbose \
    ${NON_BLOCK_FP32:+--non_block_fp32} \
    --save_dir "${SAVE_DIR:-./quantized_model}"
