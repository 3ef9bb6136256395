// Output-channel tiling shared by the MFMA kernels (gfx950 only; device code).
//
// The Cq output channels of a group are covered by MTB = Cq/16 full 16-row tiles (v_mfma_f32_16x16x4_f32, 8 passes)
// plus NSM = (Cq%16)/4 four-row blocks (v_mfma_f32_4x4x1_16B_f32, 2 passes): 16 independent 4x4 outer products, used
// as 4 k-slots x 4 pixel quads of ONE 4-channel block.  Both take the SAME B operand (lane (q,p) = channel 4j+q of
// pixel p), so the rest of a group (Cq = 24: 8 channels) costs 2 x 2 passes per k-step instead of 8 for a second,
// half-empty 16-row tile; neither M nor K carries padding.  Layouts pinned on hardware by scripts/micro/mfma4x4.hip.
#pragma once

typedef float finc_v4f __attribute__((ext_vector_type(4)));
typedef unsigned finc_v2u __attribute__((ext_vector_type(2)));

// one accumulator update: tile index mt < MTB is a 16-row tile, otherwise a 4-row block; `a` = the matching fragment
// (16-row tile: lane (q,i) = W[16mt+i][k-slot q]; 4-row block: lane (q,i) = W[16*MTB + 4sb + (i&3)][k-slot q])
template <int MTB>
__device__ inline void finc_mma(finc_v4f &acc, int mt, float a, float b)
{
    if (mt < MTB) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc, 0, 0, 0);
    else acc = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, acc, 0, 0, 0);
}

// Four 4-row-block fragments share ONE register.  A block fragment repeats its 16 values (4 rows x 4 k-slots) for each of
// the 4 pixel quads; the multi-block MFMAs can broadcast the A operand instead: with CBSZ = 2 the 16 blocks form groups
// of 4 (= the 4 pixel quads of one k-slot) and every block of a group takes A from the group's block ABID.  So lane
// (q, 4a + i) of a packed register holds fragment a's value for (row i, k-slot q), and ABID = a selects it: the 4-row
// blocks of a filter bank cost a quarter of the registers (c3: 108 -> 27), same instruction, same rate.
__device__ inline void finc_mma_small(finc_v4f &acc, float a4, float b, int abid)
{
    switch (abid & 3) {   // (the builtin wants literals; the switch folds once the loops are unrolled)
    case 0: acc = __builtin_amdgcn_mfma_f32_4x4x1f32(a4, b, acc, 2, 0, 0); break;
    case 1: acc = __builtin_amdgcn_mfma_f32_4x4x1f32(a4, b, acc, 2, 1, 0); break;
    case 2: acc = __builtin_amdgcn_mfma_f32_4x4x1f32(a4, b, acc, 2, 2, 0); break;
    default: acc = __builtin_amdgcn_mfma_f32_4x4x1f32(a4, b, acc, 2, 3, 0); break;
    }
}

// A 4-row block's register i holds, in lane row q', the k-slot-q' PARTIAL sum of channel base+i.  Sum over the 4 lane
// rows and leave channel base+q in lane row q: a 4x4 transpose-reduce (rows two apart by v_permlane32_swap + add,
// rows one apart by v_permlane16_swap + add; 6 VALU).
__device__ inline float finc_block_reduce(const finc_v4f &acc)
{
    // NB: __builtin_bit_cast applied directly to an ext-vector ELEMENT silently reads element 0 with this compiler;
    // always go through scalar temporaries.
    const float r0 = acc.x, r1 = acc.y, r2 = acc.z, r3 = acc.w;
    // v_permlane32_swap(v, s): new v = [v.lanes 0-31, s.lanes 0-31], new s = [v.lanes 32-63, s.lanes 32-63]
    const finc_v2u a = __builtin_amdgcn_permlane32_swap(__builtin_bit_cast(unsigned, r0), __builtin_bit_cast(unsigned, r2),
                                                        false, false);
    const finc_v2u b = __builtin_amdgcn_permlane32_swap(__builtin_bit_cast(unsigned, r1), __builtin_bit_cast(unsigned, r3),
                                                        false, false);
    const unsigned a0 = a.x, a1 = a.y, b0 = b.x, b1 = b.y;
    const float s02 = __builtin_bit_cast(float, a0) + __builtin_bit_cast(float, a1); // rows 0,1: r0 ; rows 2,3: r2
    const float s13 = __builtin_bit_cast(float, b0) + __builtin_bit_cast(float, b1); // rows 0,1: r1 ; rows 2,3: r3
    // v_permlane16_swap(v, s): new v = [v.row0, s.row0, v.row2, s.row2], new s = [v.row1, s.row1, v.row3, s.row3]
    const finc_v2u c = __builtin_amdgcn_permlane16_swap(__builtin_bit_cast(unsigned, s02), __builtin_bit_cast(unsigned, s13),
                                                        false, false);
    const unsigned c0 = c.x, c1 = c.y;
    return __builtin_bit_cast(float, c0) + __builtin_bit_cast(float, c1);
}

// Linv (the z-term) is lower triangular: its fragment (k-step j, tile mt) is all zero when every column 4j..4j+3 lies
// right of every row of the tile, and the MFMA can be skipped
__host__ __device__ constexpr bool finc_zterm_is_zero(int MTB, int j, int mt)
{
    return mt < MTB ? 4 * j > 16 * mt + 15 : 4 * j > 16 * MTB + 4 * (mt - MTB) + 3;
}

// row of the weight matrix held by lane i (0..15) of fragment tile mt
__host__ __device__ inline int finc_tile_row(int MTB, int mt, int i)
{
    return mt < MTB ? 16 * mt + i : 16 * MTB + 4 * (mt - MTB) + (i & 3);
}
