/*
 * hufgpu_kernels.hip - CDNA4 (gfx950) kernels of the Huffman block codec.
 *
 * One libhuffman block (config->blocksize input bytes, src/encoder.c:288-293) is the unit of
 * parallelism everywhere: blocks never exchange data, so every kernel maps blocks to
 * workgroups and the grid is simply the block count.
 *
 *   encode:  hist_tree_kernel -> pack_kernel
 *            (blocks >= 4 MiB are cut into chunks of 256 KiB, one workgroup each: chunk_hist -> block_hist ->
 *            tree -> scan_sizes -> chunk_total -> chunk_scan -> pack_chunk)
 *   decode:  decode_prepare_kernel -> decode_kernel (block index known)
 *            decode_prepare_kernel -> decode_sub_kernel -> decode_fix_kernel (block index and the
 *            encoder's sub-index known), or decode_chain_kernel (raw stream, blocks in order)
 *
 * Wave size is 64 throughout (hard-coded, gfx950 only).  All arithmetic is integer.
 */
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "hufgpu_common.h"

#include "kernels/util.hpp"
#include "kernels/histogram.hpp"
#include "kernels/tree.hpp"
#include "kernels/offsets.hpp"
#include "kernels/hist_tree.hpp"
#include "kernels/hist_lanes.hpp"
#include "kernels/pack.hpp"
#include "kernels/hist_chunk.hpp"
#include "kernels/pack_chunk.hpp"
#include "kernels/decode.hpp"
#include "kernels/decode_sub.hpp"
#include "kernels/decode_fast.hpp"
#include "kernels/decode_regs.hpp"
#include "kernels/spec_index.hpp"
#include "kernels/discover.hpp"
#include "kernels/fill.hpp"
