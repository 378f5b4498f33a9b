// ff_layout.h -- tiling shared by the host-side weight packer and the gfx950 kernels.
//
// Two f32 MFMA shapes are used (TILE = columns = samples per wavefront):
//   TILE 32: v_mfma_f32_32x32x2_f32   lane group q = lane>>5 in {0,1};   K per MFMA = 2
//   TILE 16: v_mfma_f32_16x16x4_f32   lane group q = lane>>4 in {0..3};  K per MFMA = 4
// Samples (and tangent columns) sit on the MFMA column index  col = lane & (TILE-1); features on
// the row index.  The accumulator tile holds, in register i of lane group q, row
//     TILE 32:  (i & 3) + 8 * (i >> 2) + 4 * q      (16 registers, CDNA4 C/D map)
//     TILE 16:  (i & 3) + 4 * q                       ( 4 registers)
// A vector of F features per column is F / NQ registers per lane (NQ = 64 / TILE lane groups);
// register r of group q holds feature feat(r, q) below, where RB = registers per LOGICAL BLOCK of
// 32 feature rows (16 for TILE 32; 8 for TILE 16 = two 16-row accumulator tiles).  The B operand
// of the f32 MFMA is one register per lane whose lane group q supplies k = q, so feeding
// activation register r straight back as B contracts over the features { feat(r, q) } -- the
// accumulator of one layer IS the operand of the next, with no data movement, as long as the A
// operand (the weights) is packed in the matching order.
//
// A operands are stored as CHUNKS: for (group g of 4 operand registers, logical block ob), PHYS
// sub-chunks of 1 KiB (PHYS = 32 / TILE accumulator tiles per logical block); sub-chunk p, lane l,
// float j  =  W[ 32*ob + TILE*p + (l & (TILE-1)) ][ kcol(4*g + j, l / TILE) ]
// so one 16-byte load per lane feeds 4 MFMAs.  The chunks of one evaluation form a single linear
// stream in the exact order the kernel consumes them (layer 1, hidden layers, output layer), which
// lets the kernel prefetch with one uniform sliding window across layer and evaluation
// boundaries.  Within a layer of G groups and NOB logical blocks the order is
//   phase A: group-major  -- all NOB accumulators advance together;
//   phase B (the groups of the last logical k-block, at most RB/4): block-major -- accumulator ob completes after its last
//           chunk, so its activation can run behind the MFMAs of block ob+1.
// Each layer's chunk count is padded to a multiple of kChunkPad (zero chunks, loaded but unused)
// so the prefetch ring position is a compile-time constant everywhere.
#pragma once
#include <stdint.h>
#include <stddef.h>

#if defined(__HIPCC__) || defined(__CUDACC__)
#define FF_HD __host__ __device__ inline
#else
#define FF_HD inline
#endif

namespace ff {

constexpr int kSubChunkFloats = 256;  // 1 KiB: 64 lanes x 16 bytes
constexpr int kChunkPad = 16;         // layer chunk counts are padded to a multiple of this

FF_HD constexpr int tile_nq(int tile) { return 64 / tile; }          // lane groups
FF_HD constexpr int tile_phys(int tile) { return 32 / tile; }        // accumulator tiles per logical block
FF_HD constexpr int tile_rb(int tile) { return 32 / tile_nq(tile); } // registers per logical block (16 / 8)
FF_HD constexpr int chunk_floats(int tile) { return kSubChunkFloats * tile_phys(tile); }

// feature index held by activation/state register r on lane group q
FF_HD constexpr int feat_of_reg(int tile, int r, int q)
{
    return tile == 32 ? 32 * (r >> 4) + (r & 3) + 8 * ((r & 15) >> 2) + 4 * q
                      : 32 * (r >> 3) + 16 * ((r & 7) >> 2) + 4 * q + (r & 3);
}

// number of lane registers needed for F features (multiple of 4 registers)
FF_HD constexpr int regs_for(int tile, int F)
{
    const int per4 = 4 * tile_nq(tile);      // features covered by 4 registers
    return 4 * ((F + per4 - 1) / per4);
}

// logical output blocks (32 rows) needed to produce `regs` registers
FF_HD constexpr int blocks_for_regs(int tile, int regs) { return (regs + tile_rb(tile) - 1) / tile_rb(tile); }

// Chunk order of one layer with KR operand registers and NOB logical output blocks.
struct LayerGeom {
    int G;      // groups of 4 operand registers
    int NOB;    // logical output blocks
    int GB;     // groups in phase B (block-major tail)
    int GA;     // groups in phase A (group-major)
    int NC;     // real chunks = G * NOB
    int CPAD;   // padded chunk count
};

// `gb_max` = operand-register groups of ONE logical block (RB / 4): phase B must not reach back
// into the block before the last one, because that block's activation is written in place (over
// the operands it supersedes) while phase B is still running.
FF_HD constexpr LayerGeom layer_geom(int KR, int NOB, int gb_max)
{
    LayerGeom g{};
    g.G = KR / 4;
    g.NOB = NOB;
    g.GB = g.G < gb_max ? g.G : gb_max;
    g.GA = g.G - g.GB;
    g.NC = g.G * NOB;
    g.CPAD = (g.NC + kChunkPad - 1) / kChunkPad * kChunkPad;
    return g;
}

// chunk index -> group (consumption order)
FF_HD constexpr int chunk_group(const LayerGeom& L, int c)
{
    return c < L.GA * L.NOB ? c / L.NOB : L.GA + (c - L.GA * L.NOB) % L.GB;
}
// chunk index -> logical output block
FF_HD constexpr int chunk_block(const LayerGeom& L, int c)
{
    return c < L.GA * L.NOB ? c % L.NOB : (c - L.GA * L.NOB) / L.GB;
}

// (group, logical output block) -> chunk index: the inverse of chunk_group / chunk_block
FF_HD constexpr int chunk_index(const LayerGeom& L, int g, int ob)
{
    return g < L.GA ? g * L.NOB + ob : L.GA * L.NOB + ob * L.GB + (g - L.GA);
}

struct Layout {
    int tile;     // 32 or 16
    int H;        // hidden width on chip (multiple of 32)
    int NB;       // H / 32 logical blocks
    int dregs;    // state registers
    int cregs;    // conditional registers
    int n_hidden; // hidden layers
    int nob_out;  // logical output blocks of the last layer
    LayerGeom g1, gh, go;   // first layer, hidden->hidden, output layer

    int chunks_l1, chunks_hid, chunks_out, chunks_total;   // chunk stream (in chunks)
    int chunk_fl;           // floats per chunk
    size_t stream_floats;   // chunks_total * chunk_fl
    size_t bias_floats;     // (n_hidden-1) * H + nob_out*32
    size_t total_floats;

    FF_HD int chunk_off_hid(int l) const { return chunks_l1 + l * chunks_hid; }
    FF_HD int chunk_off_out() const { return chunks_l1 + (n_hidden - 1) * chunks_hid; }
    FF_HD size_t bias_off_hid(int l) const { return stream_floats + (size_t)l * H; }
    FF_HD size_t bias_off_out() const { return stream_floats + (size_t)(n_hidden - 1) * H; }
};

FF_HD constexpr Layout make_layout(int tile, int H, int dregs, int cregs, int n_hidden)
{
    Layout L{};
    L.tile = tile; L.H = H; L.NB = H / 32; L.dregs = dregs; L.cregs = cregs; L.n_hidden = n_hidden;
    L.nob_out = blocks_for_regs(tile, dregs);
    L.g1 = layer_geom(dregs + cregs, L.NB, tile_rb(tile) / 4);
    L.gh = layer_geom(H / tile_nq(tile), L.NB, tile_rb(tile) / 4);
    L.go = layer_geom(H / tile_nq(tile), L.nob_out, tile_rb(tile) / 4);
    L.chunks_l1 = L.g1.CPAD;
    L.chunks_hid = L.gh.CPAD;
    L.chunks_out = L.go.CPAD;
    L.chunks_total = L.chunks_l1 + (n_hidden - 1) * L.chunks_hid + L.chunks_out;
    L.chunk_fl = chunk_floats(tile);
    L.stream_floats = (size_t)L.chunks_total * L.chunk_fl;
    L.bias_floats = (size_t)(n_hidden - 1) * H + (size_t)L.nob_out * 32;
    L.total_floats = L.stream_floats + L.bias_floats;
    return L;
}

// Evaluation-row header (FF_ROW_HDR = 32 four-byte words), followed by c1[H].
struct RowHdr {
    float a;            // 0  multiplies the stage state in the RHS
    float b;            // 1  multiplies the network output in the RHS
    float gn;           // 2  noise coefficient g * sqrt(|dt|)
    uint32_t flags;     // 3  FF_ROW_STEP_END | FF_ROW_NOISE
    int32_t slot;       // 4  stage slot that receives this RHS
    int32_t noise_idx;  // 5  slab of the noise array used by this row
    float pad0[2];      // 6,7
    float cin[8];       // 8..15  stage input  y = x + sum_s cin[s] * k[s]   (first 6 used)
    float cout[8];      // 16..23 step update  x += sum_s cout[s] * k[s]     (first 6 used)
    float pad1[8];      // 24..31
};
static_assert(sizeof(RowHdr) == 32 * 4, "row header is 32 words");

} // namespace ff
