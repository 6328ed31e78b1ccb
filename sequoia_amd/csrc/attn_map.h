// blockIdx -> (query head, 16-query tile) map shared by the attention kernels (csrc/tree_attention.hip, csrc/draft_block.hip):
// XCD-aware (block b runs on XCD b % 8, each XCD has its own L2), so that the workgroups of one KV head share an L2.
#pragma once
#define ATT_BM 16      // queries per workgroup

// Fewer than 8 KV heads (a tensor-parallel shard of a GQA model: 8 query heads on ONE KV head at TP = 8; tiny test
// models): every XCD that runs a workgroup of a KV head pulls that head's K/V into its own L2, so the head's
// (query head, query tile) items stay on as few XCDs as give every item a compute unit of its own (32 per XCD, one
// 512-thread workgroup each) -- never more than the head's share 8 / h_kv of the chip.  70B shard, 8:1, 129 rows:
// 72 items -> 3 XCDs instead of 8 (PMC HBM traffic of that launch was 2.74x algorithmic with the heads dealt round-robin).
static inline int att_xcd_span(int n_heads, int h_kv, int n_tiles) {
    const int budget = 8 / h_kv;                                   // >= 1 for h_kv < 8
    const int items = (n_heads / h_kv) * n_tiles;
    const int want = (items + 31) / 32;
    return want < 1 ? 1 : (want > budget ? budget : want);
}

// blockIdx.x -> (query head, query tile), or false for a block without work.  b % 8 selects the XCD (a speed hint: any
// placement is correct).  One function for the kernel and for the host-side check of the map (sq_tree_attention_block_decode:
// every (head, tile) exactly once, whatever the head counts -- tests/test_abi_and_dropin.py).
__host__ __device__ static inline bool att_decode_block(int b, int n_heads, int h_kv, int n_tiles, int xcd_span, int* head, int* tile) {
    const int xcd = b & 7, jj = b >> 3;
    const int grp = n_heads / h_kv;
    if (h_kv >= 8) {
        // GQA with enough KV heads to fill the XCDs: all query heads of a KV head share its XCD
        const int unit = grp * n_tiles;
        const int kvh_x = xcd + 8 * (jj / unit), within = jj % unit;
        if (kvh_x >= h_kv) return false;
        *head = kvh_x * grp + within / n_tiles;
        *tile = within % n_tiles;
        return true;
    }
    // KV head kvh owns XCDs [kvh * budget, kvh * budget + span); its items (query head, tile) are dealt over them
    const int budget = 8 / h_kv;
    const int kvh_x = xcd / budget, j = xcd - kvh_x * budget;
    const int item = jj * xcd_span + j;
    if (kvh_x >= h_kv || j >= xcd_span || item >= grp * n_tiles) return false;
    *head = kvh_x * grp + item / n_tiles;
    *tile = item % n_tiles;
    return true;
}

static inline int att_grid_blocks(int n_heads, int h_kv, int n_tiles, int xcd_span) {
    return h_kv >= 8 ? 8 * ((h_kv + 7) / 8) * (n_heads / h_kv) * n_tiles
                     : 8 * (((n_heads / h_kv) * n_tiles + xcd_span - 1) / xcd_span);
}

