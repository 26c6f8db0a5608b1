// pt_engine.hip -- MI355X (gfx950) path-tracing engine behind include/mi355x_pathtracer.h.
//
// Replaces the device side of the reference's src/pathtrace.cu.  Design (see DESIGN.md for the full account):
//
//  * Streams are SoA (one fp32/int32 array per field, coalesced 256-B wave accesses), not the reference's 44-B
//    PathSegment / 32-B ShadeableIntersection AoS records: 15 words per stored path (17 with texcoords).
//  * One bounce = ONE launch: the fused kernel, whose tail and whose next launch's head together are the stable multi-bin
//    partition (round 3; rounds 1-2 ran it as a second kernel, k_move):
//      k_bounce : shade(b-1) [src/pathtrace.cu:355-404 + interactions.h scatterRay] immediately followed by
//                 computeIntersections(b) [:261-344] of the scattered ray, for every path still alive; bounce 0
//                 fuses generateRayFromCamera [:206-255] instead of a shade.  Paths that end at bounce b (miss,
//                 light, last bounce, emissive texel) add their radiance to the image right there, which is
//                 what finalGather [:407-416] would do later (each pixel exactly once per iteration), and are
//                 dropped -- only paths that will scatter again are stored.
//                 Every workgroup owns a contiguous chunk of tiles and keeps running per-material counts, so a
//                 tile's prefix = (totals of earlier workgroups) + (running count inside the chunk): no scan pass.
//      the sort : the reference's stable_partition [:541] followed by the next bounce's stable sort_by_key by material [:518] is
//                 ONE order: bin (material descending), then workgroup chunk, then tile, then rank in the tile.  It is never
//                 materialised.  TAIL of k_bounce ("local move"): every workgroup sorts the stored paths of ITS chunk of tiles
//                 by (bin, tile, rank) into a chunk-local index (8 B per path: stage slot + the path's rank among all survivors of
//                 its bin inside the chunk) -- no other workgroup's data is needed for that -- and leaves the chunk's per-bin
//                 counts in a flat [bin][workgroup] "run" table.  HEAD of the next k_bounce: a workgroup that will shade sorted
//                 positions [A, B) finds the run holding A by three 64-wide scans (bins -> groups of 64 workgroups -> workgroups),
//                 keeps a window of the next 64 runs' prefixes in LDS, and every position becomes (run, offset) by a 6-step search
//                 in LDS, hence a slot of the chunk-local index, hence the record and its RNG stream index [:373] (which counts the
//                 dropped survivors too and must be the reference's).  The 60-byte records stay where k_bounce put them.
//  * The live count never visits the host: kernels read it from device memory and use grid-stride tile loops,
//    so a batch of iterations is a fixed sequence of launches (depth + 2 of them); K iterations ride in every launch as segments
//    (blockIdx.y), and three such batches are in flight on three streams so that their kernels fill each other's tails.
//  * Intersection is tile-cooperative (tileIntersect): candidate masks from conservative world boxes, the (ray, geom)
//    pairs of a 256-path tile pooled in LDS and worked off by dense waves with a 64-bit LDS minimum per ray.  Scenes
//    with BVH meshes run the mesh search as kernels of their own (k_mesh: the search, refilling waves; k_finish: the parked rays'
//    finishing) between two halves of k_bounce.
//  * Scene tables (materials, per-geom matrices, small meshes' triangles, tabulated normals) are staged in LDS; the
//    world boxes are read through the scalar cache.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <string>
#include <vector>
#include <algorithm>
#include <cmath>

#include "../../include/mi355x_pathtracer.h"
#include "pt_device.h"
#include "pt_bvh.h"

using namespace ptd;

namespace {

thread_local std::string g_last_error;
int set_error(int code, const std::string &msg) { g_last_error = msg; return code; }

#define HIPCHECK(expr)                                                                                   \
    do {                                                                                                 \
        hipError_t e_ = (expr);                                                                          \
        if (e_ != hipSuccess)                                                                            \
            return set_error(PTX_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e_));            \
    } while (0)

#ifndef PT_TILE
#define PT_TILE 256
#endif
constexpr int MAX_LANES = 8;     // launch sets in flight at most (ptx_options.lanes)
#ifndef PT_MESH_WAVES
#define PT_MESH_WAVES 5       // waves per SIMD k_mesh is compiled for
#endif
#ifndef PT_EXP_CARRY_UV
#define PT_EXP_CARRY_UV 0     // experiment only (1: the specialised kernel carries the two texcoord fields it has no use for: +8 B read per ray, +8 B written per
                              // stored path, same results -- what the wall time makes of 8 % more HBM traffic, profiles/experiments/README.md)
#endif
#ifndef PT_PARK_STATE
#define PT_PARK_STATE 1       // specialised unsplit k_bounce: state that is idle during the pair tests waits in LDS, not in registers
#endif
#ifndef PT_FAST_WAVES
#define PT_FAST_WAVES 8       // waves per SIMD the specialised k_bounce variants are compiled for (<= 64 registers; 8 workgroups'
                              // LDS is also what a CU holds with the Cornell tables since the record buffer lost a row and the window half its runs)
#endif
#ifndef PT_FAST_WAVES_FIRST
#define PT_FAST_WAVES_FIRST (PT_FAST_WAVES - 1)      // the specialised camera-ray variant: 64-66 registers.  (Compiled for 8 waves it fits 64 with two
                                    // values spilled to scratch: the kernel alone +3 %, the wall with three launch sets in flight -0.8 %, three runs
                                    // each on one box, gpurun_out/r5i_ab.log -- not taken: no kernel of the path spills, tests/test_build_resources.py)
#endif
#ifndef PT_FAST_WAVES_SPLIT
#define PT_FAST_WAVES_SPLIT 4 // same for the specialised MODE 1 variant, which carries the mesh candidate queue as well
#endif
#ifndef PT_FAST_WAVES_SPLIT2
#define PT_FAST_WAVES_SPLIT2 5 // and for MODE 2 (fits 6-7 as it is; a tighter bound measured 1-2 % slower on C5)
#endif
#ifndef PT_BOUNCE_WAVES
#define PT_BOUNCE_WAVES 4     // waves per SIMD k_bounce is compiled for (register budget 512 / this)
#endif
constexpr int TILE = PT_TILE;      // paths per tile = threads per workgroup (PT_TILE / 64 waves)
constexpr int WAVES = TILE / 64;
// a path's rank inside its tile takes RANK_BITS; the stage key packs bin | rank among all << BIN_BITS | rank among the stored << (BIN_BITS + RANK_BITS)
constexpr int RANK_BITS = TILE <= 256 ? 8 : 9, BIN_BITS = 32 - 2 * RANK_BITS;
static_assert(TILE <= 512 && (TILE & (TILE - 1)) == 0, "tile size: a power of two up to 512 (9-bit ranks)");
// words of per-tile LDS in front of the 16-byte aligned record buffer: ranking histogram, running prefix, tile counts/offsets,
// tileIntersect's 2 x 4 list counters
// + the window over the input's run tables (locate): 65 start positions, 65 stream-index bases, 64 local-index bases, next run
#ifndef PT_WIN
#define PT_WIN 32
#endif
constexpr int WIN = PT_WIN, WIN_WORDS = 2 * (WIN + 1) + WIN + 2;      // (a power of two <= 64: one wave loads a window; 32 since round 4 -- with the
                                                                 // record rows below what lets EIGHT workgroups' LDS fit a CU for the Cornell tables)
constexpr int ldsHeadWords(int nb) { return ((2 * WAVES * nb + 4 * nb + 1 + 8 + WIN_WORDS) + 3) & ~3; }
// k_bounce's dynamic LDS, in words: [scene tables][head][17 x TILE records].  The record buffer doubles as tileIntersect's
// scratch, whose 64-bit minimum keys (best[], at word 6*TILE of it) are the target of ds_min_u64: a 4-byte-misaligned
// 64-bit LDS atomic is a memory aperture violation (that is the fault of gpurun_out/bench11.log, round 1: a head of
// 2*WAVES*nb + 4*nb + 1 words put the records on an odd word, group_seg_size 20596 B = 5149 words).  Hence every part is a
// multiple of 4 words, the layout has this one definition for host and device, and the asserts below pin it.
constexpr int REC_WORDS = 17 * TILE;
// (the specialised fused kernel carries no texcoords: its pixel / material / key rows move up over one of their two, 16 rows -- what
// tileIntersect's scratch with the parked state needs anyway)
#ifndef PT_REC_ROWS_FAST0
#define PT_REC_ROWS_FAST0 16
#endif
constexpr int REC_ROWS_FAST0 = PT_REC_ROWS_FAST0;
static_assert(WIN <= 64 && (WIN & (WIN - 1)) == 0, "window of runs: one wave, binary search");
__host__ __device__ constexpr size_t bounceLdsWords(int tableWords, int nb) { return (size_t)tableWords + (size_t)ldsHeadWords(nb) + REC_WORDS; }
static_assert(ldsHeadWords(1) % 4 == 0 && ldsHeadWords(2) % 4 == 0 && ldsHeadWords(3) % 4 == 0 && ldsHeadWords(7) % 4 == 0 &&
              ldsHeadWords(45) % 4 == 0 && ldsHeadWords(65535) % 4 == 0, "record buffer must start 16-byte aligned");
static_assert(sceneTableWords(1, 1, 1) % 4 == 0 && sceneTableWords(12, 7, 7) % 4 == 0 && sceneTableWords(0, 3, 5) % 4 == 0, "scene tables end 16-byte aligned");
static_assert((6 * TILE) % 2 == 0 && (8 * TILE) % 2 == 0, "tileIntersect's 64-bit keys and lists must be 8-byte aligned inside the record buffer");

// SoA stream.  "stream" buffers hold paths waiting to be shaded (sorted); "stage" buffers hold the output of
// k_bounce in tile order; the sort exists only as the chunk-local index in their lsrc / lidx arrays plus the run tables.
struct PathSoA {
    // A stored path lies in up to FOUR arrays of 16-byte quads per slot: A = (shading point xyz, pixel slot) and B = (throughput colour rgb,
    // materialId | geomId << 16), which every record has; C = (normal xyz, texcoord u) unless the record's bin is in ntab_bins; D = (incoming
    // direction xyz, texcoord v) if its bin is in dir_bins.  One 16-byte load or store per part, and a run of a tile's records of one bin
    // is 16 B per record and array, not 4: the run's first and last cache lines, which the neighbouring bins' readers fetch as well, are
    // a fifth of what it reads instead of half (round 4: the wall time follows the HBM bytes).  The ints lie at i + k * stride (stride =
    // segments x capacity; a segment's part of an array starts seg * capacity slots further).  Kept as bases + stride rather than a pointer
    // per field: a kernel that holds two of these in scalar registers for its whole tile loop has none left for anything else.
    float *q;          // [4][stride] quads
    int32_t *i;        // [3][stride]: idx, lsrc, lidx
    uint32_t stride;
    struct F4 { float *b; __host__ __device__ float &operator[](size_t s) const { return b[s * 4]; } };      // one component of a quad array
    struct I4 { int32_t *b; __host__ __device__ int32_t &operator[](size_t s) const { return b[s * 4]; } };
    __host__ __device__ float *quadA() const { return q; }                              // px py pz pix
    __host__ __device__ float *quadB() const { return q + 4 * (size_t)stride; }         // cr cg cb mg
    __host__ __device__ float *quadC() const { return q + 8 * (size_t)stride; }         // nx ny nz u
    __host__ __device__ float *quadD() const { return q + 12 * (size_t)stride; }        // dx dy dz v
    __host__ __device__ F4 px() const { return {q}; }              // shading point = origin + t * direction (src/pathtrace.cu:392)
    __host__ __device__ F4 py() const { return {q + 1}; }
    __host__ __device__ F4 pz() const { return {q + 2}; }
    __host__ __device__ I4 pix() const { return {reinterpret_cast<int32_t *>(q) + 3}; }                          // slot among the owned pixels
    __host__ __device__ F4 cr() const { return {quadB()}; }        // throughput colour
    __host__ __device__ F4 cg() const { return {quadB() + 1}; }
    __host__ __device__ F4 cb() const { return {quadB() + 2}; }
    __host__ __device__ I4 mg() const { return {reinterpret_cast<int32_t *>(quadB()) + 3}; }                     // materialId | geomId << 16
    __host__ __device__ F4 nx() const { return {quadC()}; }        // pending intersection: normal, texcoords (u, v only if textured)
    __host__ __device__ F4 ny() const { return {quadC() + 1}; }
    __host__ __device__ F4 nz() const { return {quadC() + 2}; }
    __host__ __device__ F4 u() const { return {quadC() + 3}; }
    __host__ __device__ F4 dx() const { return {quadD()}; }        // incoming direction
    __host__ __device__ F4 dy() const { return {quadD() + 1}; }
    __host__ __device__ F4 dz() const { return {quadD() + 2}; }
    __host__ __device__ F4 v() const { return {quadD() + 3}; }
    // logical field k of slot j, in the order the debug capture hands them out: point, direction, colour, normal, u, v
    __host__ __device__ float fieldAt(int k, size_t j) const {
        if (k < 3) return q[j * 4 + k];
        if (k < 6) return quadD()[j * 4 + (k - 3)];
        if (k < 9) return quadB()[j * 4 + (k - 6)];
        if (k < 12) return quadC()[j * 4 + (k - 9)];
        return k == 12 ? quadC()[j * 4 + 3] : quadD()[j * 4 + 3];
    }
    __host__ __device__ int32_t *idx() const { return i; }     // stage key (see stage_key) or -1
    // the chunk-local sorted index a workgroup leaves in its tail (see "local move" in k_bounce): entry e of the chunk's region is
    // the stage slot of the path that comes e-th in (bin, tile, rank) order inside the chunk, and its rank among ALL survivors of
    // its bin inside the chunk (the part of the RNG stream index the workgroup can know by itself)
    __host__ __device__ int32_t *lsrc() const { return i + (size_t)stride; }
    __host__ __device__ int32_t *lidx() const { return i + 2 * (size_t)stride; }
};
constexpr int SOA_FLOATS = 16, SOA_INTS = 3, SOA_LOGICAL_FLOATS = 14;      // words per slot in the float / int buffers; fields the capture hands out

// stage key: bin | rank among all survivors of the tile << 16 | rank among the stored ones << 24 (ranks < 256)
__device__ __forceinline__ int32_t stage_key(int bin, int r_all, int r_scat) { return (int32_t)((uint32_t)bin | ((uint32_t)r_all << BIN_BITS) | ((uint32_t)r_scat << (BIN_BITS + RANK_BITS))); }

__device__ __forceinline__ PathSoA soa_offset(PathSoA s, size_t off) {
    s.q += 4 * off; s.i += off;
    return s;
}
// The same stream, but with a stride the optimiser cannot see through: field addresses derived from the result are
// computed where they are used (a few scalar adds per tile) instead of being hoisted out of the tile loop and kept --
// 34 scalar registers per stream -- for its whole length.
// Uniform base + per-lane 32-bit byte offset: the form the hardware addresses by itself (global_load_dword v, voffset,
// s[base:base+1]).  Written as base[index] the compiler forms a 64-bit address per access in vector registers (one
// v_lshl_add_u64 and a register pair each); this way a record's 15 fields share one offset register and the per-field
// bases are scalar adds.  The base must be wave-uniform and the offset below 4 GiB (a segment's field is capacity x 4 B).
#ifndef PT_SCALAR_BASE
#define PT_SCALAR_BASE 1
#endif
template <class T> using gptr = T __attribute__((address_space(1))) *;
template <class T> __device__ __forceinline__ T ld_u(const T *base, uint32_t byteoff) {
#if PT_SCALAR_BASE
    gptr<const T> b = (gptr<const T>)base;
    asm volatile("" : "+s"(b));
    return *reinterpret_cast<gptr<const T>>(reinterpret_cast<gptr<const char>>(b) + byteoff);
#else
    return *reinterpret_cast<const T *>(reinterpret_cast<const char *>(base) + byteoff);
#endif
}
template <class T> __device__ __forceinline__ void st_u(T *base, uint32_t byteoff, T v) {
#if PT_SCALAR_BASE
    gptr<T> b = (gptr<T>)base;
    asm volatile("" : "+s"(b));
    *reinterpret_cast<gptr<T>>(reinterpret_cast<gptr<char>>(b) + byteoff) = v;
#else
    *reinterpret_cast<T *>(reinterpret_cast<char *>(base) + byteoff) = v;
#endif
}

__device__ __forceinline__ PathSoA soa_fresh(PathSoA s) {
    asm volatile("" : "+s"(s.stride));
    return s;
}

struct TileMap {           // which pixels this device owns (row blocks round-robin over tile_world)
    int32_t W, H, tile_rows, tile_rank, tile_world, owned;
    uint32_t w_mul, w_sh, rows_mul, rows_sh;     // n / W and n / tile_rows as multiply-high + shift (fastdiv), set by the host
};

// n / d for 0 <= n < 2^31 and a divisor fixed at create: q = (n * mul) >> (32 + sh) with mul = floor(2^(32+sh) / d) + 1,
// sh = ceil(log2 d) - 1 (exact for that range: the classic invariant-divisor multiply); sh = 255 marks d == 1.
// Keeps the compiler's generic division -- a dozen instructions and a loop-invariant reciprocal that it spills -- out of
// the tile loop.
__device__ __forceinline__ int fastdiv(int n, uint32_t mul, uint32_t sh) {
    return sh == 255u ? n : (int)(__umulhi((uint32_t)n, mul) >> sh);
}

__device__ __forceinline__ void owned_pixel(const TileMap &tm, int i, int &x, int &y) {
    int r = fastdiv(i, tm.w_mul, tm.w_sh);
    x = i - r * tm.W;
    if (tm.tile_world <= 1) { y = r; return; }
    int k = fastdiv(r, tm.rows_mul, tm.rows_sh);
    y = (k * tm.tile_world + tm.tile_rank) * tm.tile_rows + (r - k * tm.tile_rows);
}

// global pixel index (x + y*W) of the `slot`-th pixel this device owns.  Paths carry the SLOT, not the pixel: with a
// row-tile split the per-iteration radiance buffers, like the streams, are then sized and indexed by what the device
// owns (1/8 of the frame on one of eight ranks), and with one device slot == pixel.
__device__ __forceinline__ int slot_to_pixel(const TileMap &tm, int slot) {
    if (tm.tile_world <= 1) return slot;
    int x, y;
    owned_pixel(tm, slot, x, y);
    return x + y * tm.W;
}

struct BounceParams {
    DScene sc;
    DCamera cam;
    TileMap tm;
    PathSoA in, stage;                     // in = the stage the previous bounce wrote (tile order) with its chunk-local sorted index
    // the run tables of the launch that wrote `in` (its grid had in_gx workgroups per segment; run r = bin * in_gx + workgroup):
    const int32_t *in_totals;              // [2][nbins]: survivors / stored paths per bin (n_in = sum of the stored ones)
    const int32_t *in_super;               // [2][nbins][nsuper]: the same per 64 consecutive workgroups
    const int32_t *in_chunk;               // [3][chunk_cap]: per run -- survivors, stored paths, start of the run in the local index
    int32_t in_gx;
    size_t seg_in_totals, seg_in_chunk;    // per-segment strides of those (0: the cached bounce 0, shared by all segments)
    float *image;
    int32_t iter, traceDepth, bounce;      // bounce = index b of the intersect stage done by this launch
    int32_t iter_stride;                   // iteration of segment s = iter + s * iter_stride (1; world size when ranks take turns)
    // split mesh search (MODE 1 / 2 of k_bounce, k_mesh in between): per-ray keys, the queue of parked rays (their stage slots)
    unsigned long long *keys; uint32_t *items; int32_t *item_count;
    int32_t *item_cursor;                  // one int per segment, after the counts: where k_mesh's waves draw their next chunk of the queue
    size_t seg_keys, seg_items;            // per-segment strides of keys / items; item_count has one int per segment
    const uint32_t *tile_geoms;            // first bounce: [tile] bit g = some pixel of the tile may see geom g (NULL: no information)
    int32_t *tile_done;                    // split first bounce: [segment][tile] 1 = pass 1 finished the tile (no ray of it reaches a mesh's box)
    int32_t aa, dof, sort;
    int32_t uses_uv;                       // some OBJ geom has a texture: texcoords are carried, otherwise not
    unsigned long long dir_bins;           // bit b: the records of material bin b carry the incoming direction (reflective, refractive, or a
                                           // material of an OBJ geom: what scatterRay reads it for); the other bins' records do not
    unsigned long long in_dir_bins, in_ntab_bins;      // dir_bins / ntab_bins of the launch that wrote `in` (the same, unless `in` is the cached camera bounce)
    // The local index as ONE word per stored path (round 5), where a workgroup's chunk of the writing launch is at most 128 tiles (32 768
    // slots; the host knows the bound: ceil(maxTiles / workgroups)): entry e = (slot - e) as 16 signed bits | the path's rank in its run
    // << 16 -- slot and entry lie in the same chunk's region, so their distance fits, and so does a rank below the chunk's slots.  Half
    // the index bytes: 4 B less read per ray, 4 B less written per stored path.  Larger chunks (8K frames with few iterations per set)
    // keep the two words, lsrc and lidx.  idx16: what THIS launch's tail writes; in_idx16: what the launch that wrote `in` did.
    int32_t idx16, in_idx16;
    unsigned long long ntab_bins;          // bit b: every hit of material bin b is a cube hit (no sphere or OBJ geom has the material): its records
                                           // carry the 3-bit code of the cube's tabulated normal in pix's bits 28-30 instead of the normal
    int32_t apps;                          // apps/src variant: radiance * PI at gather, albedo AOV on iteration 1
    float *albedo;
    int32_t nbins, maxTiles;
    int32_t *counts_all, *counts_scat;     // [nbins][maxTiles]: prefix of the tile inside its workgroup's chunk
    int32_t *chunk;                        // out, [3][chunk_cap], run r = bin * gridDim.x + workgroup: survivors and stored paths of
                                           // that workgroup's chunk of tiles in that bin, and where the run starts in the local index
    int32_t chunk_cap;                     // nbins x (workgroups per segment at most)
    int32_t *super_all, *super_scat;       // [nbins][nsuper]:   totals per 64 consecutive workgroups (atomics)
    int32_t *totals_all, *totals_scat;     // [nbins] of this bounce (atomics)
    int32_t nsuper;
    // batching: blockIdx.y = segment = one iteration of the batch (iteration p.iter + segment), each an independent
    // stream with its own buffers at these strides (in elements)
    size_t seg_in, seg_stage, seg_counts, seg_chunk, seg_totals, seg_part;
    unsigned long long *stamps;            // diagnostic build (-DPT_STAMPS) only: cycles per phase, summed over waves
    float *part;                           // != NULL: every ending path STORES its radiance to part[segment][pix]
                                           // (k_gather adds the segments to the image in iteration order)
    // first-bounce cache fill (iter 1, AA and DoF off): bounce-0 light hits are replayed on later iterations
    int32_t *emit_count; int32_t *emit_pix; float *emit_rgb;
    // Fences that report.  Every index the kernels take from a table another launch wrote -- a queue entry of the split mesh search,
    // a parked ray's owner, an entry of the local index, a sorted position's place in it -- is checked against `fence_slots` (the
    // stage's capacity, maxTiles x TILE) before it becomes an address: a bad one is skipped or clamped, so it costs a wrong pixel and
    // not a fault, and is COUNTED here (ptx_stats.fenced, 0 in every test): a wrong pixel is never the only symptom.
    unsigned long long *fenced;
    uint32_t fence_slots;
    uint32_t fence_slots_cap;              // maxTiles x TILE itself (fence_slots is that too, unless a test lowered it): where a segment's lit flags start
};
__device__ __forceinline__ void fence_report(const BounceParams &p) { atomicAdd(p.fenced, 1ull); }

__device__ __forceinline__ int sum_totals(const int32_t *t, int n) {
    int s = 0;
    for (int b = 0; b < n; b++) s += t[b];
    return s;
}

// A path that ends adds its radiance to its pixel (finalGather, src/pathtrace.cu:407-416).  Each pixel ends exactly
// once per iteration, so this is a plain read-modify-write, or -- when several iterations are in flight as
// segments of one launch -- a plain store into that iteration's buffer.
// three floats stored by one instruction (global_store_dwordx3 with a scalar base); the address is only 4-byte aligned
typedef float Rgb __attribute__((ext_vector_type(3)));
typedef Rgb RgbUnaligned __attribute__((aligned(4)));
__device__ __forceinline__ void st_rgb(float *base, uint32_t byteoff, float r, float g, float b) {
    const Rgb v = {r, g, b};
#if PT_SCALAR_BASE
    gptr<float> sb = (gptr<float>)base;
    asm volatile("" : "+s"(sb));
    *reinterpret_cast<RgbUnaligned __attribute__((address_space(1))) *>(reinterpret_cast<gptr<char>>(sb) + byteoff) = v;
#else
    *reinterpret_cast<RgbUnaligned *>(reinterpret_cast<char *>(base) + byteoff) = v;
#endif
}
// Batched mode, round 5: only paths that end WITH radiance (a light hit, an emissive texel) write their 12 bytes, and set the pixel's byte
// in the segment's "lit" flags, which lie behind the segment's radiance ([cap] floats x 3, then [cap] bytes; cleared per batch, 1 B per
// pixel).  The many that end black -- misses, the last bounce: most path ends of a Cornell frame -- write nothing, and k_gather adds only
// flagged slots: adding the +0 they used to store changes no sum (the image holds no -0: it starts at +0 and only grows), so the frames
// are the same bits.  C4 moved 12 B per path end and 12 B per pixel and iteration in k_gather for those zeros: 11 % of its HBM bytes.
__device__ __forceinline__ uint8_t *lit_flags(float *part, uint32_t cap) { return reinterpret_cast<uint8_t *>(part + 3 * (size_t)cap); }
__device__ __forceinline__ void deposit(const TileMap &tm, float *image, float *part, bool batched, int pix, vec3 c, int apps, uint32_t cap) {
    if (apps) c = scale(c, 3.14159265358f);            // apps/src/pathtrace.cu:44,508: image += color * PI
    if (batched) {
        st_rgb(part, (uint32_t)pix * 12u, c.x, c.y, c.z);
        st_u(lit_flags(part, cap), (uint32_t)pix, (uint8_t)1);
    } else {
        float *px = image + (size_t)slot_to_pixel(tm, pix) * 3;
        px[0] += c.x; px[1] += c.y; px[2] += c.z;
    }
}

// Albedo AOV of the apps/src copy (apps/src/pathtrace.cu:412-462): what the first hit of iteration 1 looks like.
__device__ __forceinline__ void write_albedo(const DScene &sc, const Hit &hit, float *dst) {
    vec3 a = V3(0.f, 0.f, 0.f);
    if (hit.t > 0.0f) {
        const DMaterial m = getMaterial(sc, hit.mat);
        const DGeom &geom = sc.geoms[hit.geom];
        a = V3(m.color[0], m.color[1], m.color[2]);
        if (geom.type == G_OBJ) {
            const DTex &kd = geom.tex[0], &ke = geom.tex[2];
            vec3 emission = V3(0.f, 0.f, 0.f);
            if (ke.ch) {
                int pixelID = (int)(hit.v * ke.h) * ke.w + (int)(hit.u * ke.w);
                emission = V3(texel(sc, ke, pixelID, 0) / 255.f, texel(sc, ke, pixelID, 1) / 255.f, texel(sc, ke, pixelID, 2) / 255.f);
            }
            const float eps = 1.1920928955078125e-07f;
            if (emission.x > eps || emission.y > eps || emission.z > eps) a = scale(emission, 5.0f);
            else if (kd.ch) {
                int pixelID = (int)(hit.v * kd.h) * kd.w + (int)(hit.u * kd.w);
                a = V3(texel(sc, kd, pixelID, 0) / 255.f, texel(sc, kd, pixelID, 1) / 255.f, texel(sc, kd, pixelID, 2) / 255.f);
            }
        } else if (m.emittance > 0.0f) a = scale(a, m.emittance);
        else if (m.hasRefractive > 0.0f) a = V3(m.speccolor[0], m.speccolor[1], m.speccolor[2]);
    }
    dst[0] = a.x; dst[1] = a.y; dst[2] = a.z;
}

// number of set bits of a wave ballot below this lane
__device__ __forceinline__ int wavePrefix(unsigned long long b, int lane) {
    return (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(b >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)b, 0u));
}

// computeIntersections for a whole tile, cooperatively.  Every ray lists the geoms whose conservative world box it
// reaches (cullMask); the (ray, geom) pairs of the tile are pooled in LDS -- cubes and spheres first, meshes after --
// and worked off by dense waves, each pair folding its result into its ray's 64-bit minimum with an LDS atomic
// (primKey / meshKey / packKey: min key = nearest t, lowest geom index on ties, i.e. the reference's answer).  In a
// wave of incoherent rays this replaces "every lane waits for all 7 geoms" by "about 1.3 pairs per ray, packed".
// Must be called by all threads of the workgroup (barriers inside); `scratch` = TILE*17 words of LDS.
constexpr int ITEMS_PER_PASS = 4;                      // pairs a ray may contribute per pass (1024-entry list)
#ifdef PT_STAMPS
#define TI_STAMP(k) do { unsigned long long t1_ = __builtin_amdgcn_s_memtime(); st_acc[k] += t1_ - st_t0; st_t0 = t1_; } while (0)
#define TI_ARGS , unsigned long long *st_acc, unsigned long long &st_t0
#define TI_PASS , st_acc, st_t0
#else
#define TI_STAMP(k) do { } while (0)
#define TI_ARGS
#define TI_PASS
#endif
// DEFER: the mesh pairs are not worked off here; the caller gets the best key over cubes and spheres and the ray's
// mesh candidates (split mesh search, see k_mesh), and `hit` is left alone.
template <bool DEFER, bool PARK = false, bool SUBSET = false>
__device__ __forceinline__ void tileIntersect(const DScene &sc, bool alive, Ray ray, bool need_uv, Hit &hit, int32_t *scratch,
                                              int32_t *tcnt, int &q, int tid, int lane, int wave, unsigned long long &key_out,
                                              uint32_t &mesh_out TI_ARGS, uint32_t subset = 0xffffffffu) {
    const float *gtab = reinterpret_cast<const float *>(pt_lds) + sc.ntri_lds * 24 + sc.nmats * 11;
    float *rayb = reinterpret_cast<float *>(scratch);                              // [6][TILE]
    unsigned long long *best = reinterpret_cast<unsigned long long *>(scratch + 6 * TILE);   // [TILE]
    uint16_t *list = reinterpret_cast<uint16_t *>(scratch + 8 * TILE);             // [CAP] ray | geom << RANK_BITS: cubes from the front,
    uint16_t *listM = list + ITEMS_PER_PASS * TILE;                                // spheres from the back; [CAP] meshes
    constexpr int CAP = ITEMS_PER_PASS * TILE;
    uint32_t cube_mask = 0, sph_mask = 0, mesh_mask = 0;
    if (alive) {
        const uint32_t m = cullMask<SUBSET>(sc, ray, subset);
        cube_mask = m & sc.cube_bits; sph_mask = m & sc.sphere_bits; mesh_mask = m & sc.mesh_bits;
    }
    mesh_out = mesh_mask;
    if (DEFER) mesh_mask = 0;
    rayb[0 * TILE + tid] = ray.o.x; rayb[1 * TILE + tid] = ray.o.y; rayb[2 * TILE + tid] = ray.o.z;
    rayb[3 * TILE + tid] = ray.d.x; rayb[4 * TILE + tid] = ray.d.y; rayb[5 * TILE + tid] = ray.d.z;
    best[tid] = KEY_NONE;
    for (;;) {
        // this pass: up to ITEMS_PER_PASS pairs per ray -- cubes, then spheres, then meshes, each kind in a run of
        // its own so that the waves working the list off run one kind of test.  Slots: prefix inside the wave from
        // ballots of the 3-bit counts, one LDS atomic per wave and kind for its base (tcnt[4q..]: cube, sphere and
        // mesh pairs, "some ray has more"; the other parity's counters are cleared meanwhile for the next pass).
        const int cc = (int)__popc(cube_mask), cs = (int)__popc(sph_mask), cm = (int)__popc(mesh_mask);
        const int nc = cc < ITEMS_PER_PASS ? cc : ITEMS_PER_PASS;
        const int ns = cs < ITEMS_PER_PASS - nc ? cs : ITEMS_PER_PASS - nc;
        const int nm = cm < ITEMS_PER_PASS - nc - ns ? cm : ITEMS_PER_PASS - nc - ns;
        int base[3], tot[3];
        const int cnt3[3] = {nc, ns, nm};
#pragma unroll
        for (int kind = 0; kind < 3; kind++) {
            const unsigned long long b0 = __ballot(cnt3[kind] & 1), b1 = __ballot(cnt3[kind] & 2), b2 = __ballot(cnt3[kind] & 4);
            base[kind] = wavePrefix(b0, lane) + 2 * wavePrefix(b1, lane) + 4 * wavePrefix(b2, lane);
            tot[kind] = __popcll(b0) + 2 * __popcll(b1) + 4 * __popcll(b2);
        }
        const unsigned long long left = __ballot(cc + cs + cm > nc + ns + nm);
        int wb0 = 0, wb1 = 0, wb2 = 0;
        if (lane == 0) {
            if (tot[0]) wb0 = atomicAdd(&tcnt[4 * q + 0], tot[0]);
            if (tot[1]) wb1 = atomicAdd(&tcnt[4 * q + 1], tot[1]);
            if (tot[2]) wb2 = atomicAdd(&tcnt[4 * q + 2], tot[2]);
            if (left) tcnt[4 * q + 3] = 1;
        }
        base[0] += __builtin_amdgcn_readfirstlane(wb0);
        base[1] += __builtin_amdgcn_readfirstlane(wb1);
        base[2] += __builtin_amdgcn_readfirstlane(wb2);
        for (int j = 0; j < nc; j++) {
            const int g = __ffs((int)cube_mask) - 1;
            cube_mask &= cube_mask - 1;
            list[base[0] + j] = (uint16_t)(tid | (g << RANK_BITS));
        }
        for (int j = 0; j < ns; j++) {
            const int g = __ffs((int)sph_mask) - 1;
            sph_mask &= sph_mask - 1;
            list[CAP - 1 - (base[1] + j)] = (uint16_t)(tid | (g << RANK_BITS));
        }
        for (int j = 0; j < nm; j++) {
            const int g = __ffs((int)mesh_mask) - 1;
            mesh_mask &= mesh_mask - 1;
            listM[base[2] + j] = (uint16_t)(tid | (g << RANK_BITS));
        }
        __syncthreads();
        const int totC = tcnt[4 * q + 0], totS = tcnt[4 * q + 1], totM = tcnt[4 * q + 2], more = tcnt[4 * q + 3];
        if (tid < 4) tcnt[4 * (q ^ 1) + tid] = 0;
        TI_STAMP(5);
#ifdef PT_STAMPS
        if (tid == 0) { st_acc[8] += totC + totS; st_acc[9] += totM; st_acc[10] += 1; }
#endif
        // each kind starts on a wave boundary: cubes [0, totC), spheres from roundup64(totC), meshes after them
        // small meshes (no BVH in the scene): every (ray, mesh) pair becomes mesh_chunks entries, one per group of
        // MESH_CHUNK faces, chunk-major so that a wave reads the same faces
        const int startS = (totC + 63) & ~63, startM = startS + ((totS + 63) & ~63);
        const int nch = sc.mesh_chunks > 1 ? sc.mesh_chunks : 1;
        for (int k = tid; k < startM + totM * nch; k += TILE) {
            int item = -1, chunk = -1;
            if (k < totC) item = list[k];
            else if (k >= startS && k < startS + totS) item = list[CAP - 1 - (k - startS)];
            else if (k >= startM) {
                int kk = k - startM;
                if (nch > 1) { chunk = 0; while (kk >= totM) { kk -= totM; chunk++; } }
                item = listM[kk];
            }
            if (item >= 0) {
                const int src = item & (TILE - 1), g = item >> RANK_BITS;
                Ray r;
                r.o = V3(rayb[0 * TILE + src], rayb[1 * TILE + src], rayb[2 * TILE + src]);
                r.d = V3(rayb[3 * TILE + src], rayb[4 * TILE + src], rayb[5 * TILE + src]);
                // (tileIntersect runs with the tables staged; the triangle tables too unless the scene's meshes are too big)
                // (DEFER: the mesh list stays empty, so its tests are not compiled into that kernel at all)
                const unsigned long long key = (DEFER || k < startM) ? primKey(gtab, g, r)
                    : ((sc.tri_lds == 2 || sc.ntri_lds) ? meshKey<true>(sc, gtab, g, r, chunk) : meshKey<false>(sc, gtab, g, r, chunk));
                if (key != KEY_NONE) atomicMin(&best[src], key);
            }
        }
        __syncthreads();
        q ^= 1;
        TI_STAMP(6);
        // another pass only if some ray still has candidates (rare: more than ITEMS_PER_PASS boxes along one ray)
        if (!more) break;
    }
    // no barrier here: the caller passes at least two before it touches `scratch` again
    key_out = best[tid];
    if (PARK) {                                    // the ray was not kept in registers across the pair tests: take the LDS copy
        asm volatile("" ::: "memory");
        ray.o = V3(rayb[0 * TILE + tid], rayb[1 * TILE + tid], rayb[2 * TILE + tid]);
        ray.d = V3(rayb[3 * TILE + tid], rayb[4 * TILE + tid], rayb[5 * TILE + tid]);
    }
    if (!DEFER) decodeKey(sc, gtab, key_out, ray, need_uv, hit);
    TI_STAMP(7);
}

// What pass 1 of the split bounce (and k_finish, for the rays pass 1 parked) tells pass 2 about ray i: one word, lsrc[i] of the stage
// until the tail overwrites it.  CAND = parked with mesh candidates in slot (bits 16-23) of its tile, k_finish will replace the
// word; otherwise the ray is finished -- alive / stored flags, its bin (bits 0-15) and, if stored, the slot its record lies in.
constexpr int32_t K1_CAND = (int32_t)0x80000000u, K1_ALIVE = 0x40000000, K1_PEND = 0x20000000;

// The terminal cases of shadeFakeMaterial(b) for a path whose nearest hit is known (src/pathtrace.cu:380-390, :400): light =>
// radiance, miss or last bounce => black, otherwise the path is stored for the next bounce (pending); also the path's material bin.
template <bool FIRST>
__device__ __forceinline__ void classifyPath(const BounceParams &p, int iter, float *part, bool batched, const Hit &hit, vec3 color, int pix,
                                             int &bin, bool &pending) {
    bin = p.sort ? (p.sc.nmats - 1 - hit.mat) : 0;           // material descending; a miss carries id 0
    if (FIRST && p.albedo && iter == 1) write_albedo(p.sc, hit, p.albedo + (size_t)slot_to_pixel(p.tm, pix) * 3);
    bool lit = false;
    if (hit.t > 0.0f) {
        const DMaterial m = getMaterial(p.sc, hit.mat);
        if (m.emittance > 0.0f) {                           // src/pathtrace.cu:380-383
            lit = true;
            vec3 c = mul(color, scale(V3(m.color[0], m.color[1], m.color[2]), m.emittance));
            deposit(p.tm, p.image, part, batched, pix, c, p.apps, p.fence_slots_cap);
            if (FIRST && p.emit_count) {
                const vec3 cd = p.apps ? scale(c, 3.14159265358f) : c;
                int k = atomicAdd(p.emit_count, 1);
                p.emit_pix[k] = pix;
                p.emit_rgb[k * 3 + 0] = cd.x; p.emit_rgb[k * 3 + 1] = cd.y; p.emit_rgb[k * 3 + 2] = cd.z;
            }
        } else if (p.traceDepth - p.bounce != 1) {         // :387-390 (last bounce => black)
            pending = true;
        }
    }
    // a miss or a last-bounce hit ends the path with colour 0 (:388, :400): nothing to add to the image -- and, since round 5, nothing to
    // store in batched mode either (the pixel's "lit" flag of this iteration stays clear: k_gather skips the slot)
    (void)lit;
}

// inclusive prefix sum over the lanes of a wave, in the vector ALU's own lane network (DPP): four shifted adds inside the rows of 16
// lanes, then lane 15 of a row broadcast to the next row (rows 1 and 3) and lane 31 to the upper half -- six dependent vector
// instructions.  (__shfl_up goes through the LDS crossbar: six ds_bpermute round trips, 0.3 us on a tile's critical path each time.)
__device__ __forceinline__ int waveInclusiveScan(int v, int lane) {
    (void)lane;
    v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, false);      // row_shr:1 (lanes without a source add 0)
    v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, false);      // row_shr:2
    v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xf, false);      // row_shr:4
    v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xf, false);      // row_shr:8
    v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xa, 0xf, false);      // row_bcast:15 into rows 1 and 3
    v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xc, 0xf, false);      // row_bcast:31 into rows 2 and 3
    return v;
}

// ---- where a sorted position lies (head of the sort, see the file comment) --------------------------------------------------
// The previous bounce left its stored paths as RUNS: run r = bin * gx + workgroup holds cs[r] stored paths (of ca[r] survivors)
// and starts at cbase[r] in the chunk-local index; in run order (bin-major) the runs ARE the sorted stream.  Position P of that
// stream lies in the first run whose inclusive prefix of cs exceeds P.
//
// scanFind: one wave walks n entries of (stored, survivors) counts 64 at a time, adding them to the running prefixes pre_s /
// pre_a, and stops at the first entry whose inclusive stored-prefix exceeds A; returns its index (-1: none), with pre_s / pre_a
// the prefixes in FRONT of it.  All arguments and results are wave-uniform.
__device__ __forceinline__ int scanFind(const int32_t *cs, const int32_t *ca, int n, int A, int &pre_s, int &pre_a, int lane) {
    for (int base = 0; base < n; base += 64) {
        const int k = base + lane;
        const int vs = k < n ? cs[k] : 0, va = k < n ? ca[k] : 0;
        const int is = waveInclusiveScan(vs, lane), ia = waveInclusiveScan(va, lane);
        const unsigned long long m = __ballot(k < n && pre_s + is > A);
        if (m) {
            const int l = __ffsll((long long)m) - 1;
            pre_s += __builtin_amdgcn_readlane(is - vs, l);
            pre_a += __builtin_amdgcn_readlane(ia - va, l);
            return base + l;
        }
        pre_s += __builtin_amdgcn_readlane(is, 63);
        pre_a += __builtin_amdgcn_readlane(ia, 63);
    }
    return -1;
}
// The same search with EIGHT consecutive entries per lane: 512 entries per memory round trip instead of 64 (the loads of a trip are
// all requested before the first is used).  The tables searched this way are short (bins x groups of 64 workgroups).
__device__ __forceinline__ int scanFind8(const int32_t *cs, const int32_t *ca, int n, int A, int &pre_s, int &pre_a, int lane) {
    for (int base = 0; base < n; base += 512) {
        int vs[8], va[8];
        const int k0 = base + lane * 8;
#pragma unroll
        for (int j = 0; j < 8; j++) { vs[j] = k0 + j < n ? cs[k0 + j] : 0; va[j] = k0 + j < n ? ca[k0 + j] : 0; }
        int ss = 0, sa = 0;
#pragma unroll
        for (int j = 0; j < 8; j++) { ss += vs[j]; sa += va[j]; }
        const int is = waveInclusiveScan(ss, lane), ia = waveInclusiveScan(sa, lane);
        const unsigned long long m = __ballot(pre_s + is > A);
        if (m) {
            const int l = __ffsll((long long)m) - 1;                 // the lane whose eight entries hold the answer
            pre_s += __builtin_amdgcn_readlane(is - ss, l);
            pre_a += __builtin_amdgcn_readlane(ia - sa, l);
            int found = -1;
#pragma unroll
            for (int j = 0; j < 8; j++) {                            // (uniform: every lane walks lane l's entries)
                const int es = __builtin_amdgcn_readlane(vs[j], l), ea = __builtin_amdgcn_readlane(va[j], l);
                if (found < 0) {
                    if (pre_s + es > A) found = base + l * 8 + j;
                    else { pre_s += es; pre_a += ea; }
                }
            }
            return found;
        }
        pre_s += __builtin_amdgcn_readlane(is, 63);
        pre_a += __builtin_amdgcn_readlane(ia, 63);
    }
    return -1;
}

// The window: runs [r0, r0 + WIN) of the table with their exclusive prefixes, in LDS (one wave; the caller brackets it with
// barriers).  win[0 .. WIN] = first sorted position of each run and of what follows the window, win[WIN+1 .. 2*WIN+1] = the same
// for the survivor counts (= the RNG stream index of a run's first survivor), then the runs' starts in the local index, then r0 + WIN.
__device__ __forceinline__ void windowLoad(int32_t *win, const int32_t *chunk, int chunk_cap, int nruns, int r0, int gs0, int ga0, int lane) {
    const int r = r0 + lane;
    int ca = 0, cs = 0, cb = 0;
    if (r < nruns && lane < WIN) { ca = chunk[r]; cs = chunk[chunk_cap + r]; cb = chunk[2 * chunk_cap + r]; }
    const int is = waveInclusiveScan(cs, lane), ia = waveInclusiveScan(ca, lane);
    if (lane < WIN) {
        win[lane] = gs0 + is - cs;
        win[WIN + 1 + lane] = ga0 + ia - ca;
        win[2 * (WIN + 1) + lane] = cb;
    }
    if (lane == WIN - 1) { win[WIN] = gs0 + is; win[2 * WIN + 1] = ga0 + ia; win[2 * (WIN + 1) + WIN] = r0 + WIN; }
}

// One bounce.  FIRST: generate camera rays; otherwise shade the stored paths of the previous bounce.
#ifndef PT_QCAP
#define PT_QCAP (4 * TILE)
#endif
#ifndef PT_DIRECT_STORE
#define PT_DIRECT_STORE 1      // round 5: a tile's stored paths go from registers to their stage slots (no transposition through LDS), every wave
                               // derives the tile's in-tile offsets itself, keys are written for stored slots only: three barriers per tile instead of six
#endif
#ifndef PT_RANK_ONE_BARRIER
#define PT_RANK_ONE_BARRIER 0  // experiment (see RANK1 in k_bounce): the ranking pass of the split bounce with one barrier per tile
#endif
#ifndef PT_RANK_SLICED
#define PT_RANK_SLICED 1       // later bounces, <= 16 bins: the in-wave ranking bit-sliced instead of one pass per bin that occurs
#endif
constexpr int QCAP = PT_QCAP;                        // LDS queue entries of MODE 1, behind the record buffer, + its two counters
constexpr int QUEUE_WORDS = QCAP + 4;
// MODE 1: LDS queue -> global queue of the segment (all threads of the workgroup; uniform call)
__device__ __forceinline__ void flushQueue(const BounceParams &p, int seg, const uint32_t *qbuf, int32_t *qcnt, int32_t *qbase, int tid) {
    const int n = *qcnt;
    if (tid == 0) *qbase = atomicAdd(p.item_count + seg, n);
    __syncthreads();
    uint32_t *dst = p.items + p.seg_items * seg + *qbase;
    for (int k = tid; k < n; k += TILE) dst[k] = qbuf[k];
    __syncthreads();
    if (tid == 0) *qcnt = 0;
    __syncthreads();
}

// MODE 0: the whole bounce.  Scenes with BVH meshes split it so that the mesh search -- few rays of a tile, each a long
// chain of dependent node visits -- does not hold the tile's other waves at a barrier: MODE 1 does the whole bounce for the
// rays that reach no mesh's box and, for the others, everything up to the best hit among cubes and spheres; those it parks
// (origin, direction, colour, pixel, candidate mask in a stage slot at the top of the tile, the key in `keys`) with one queue
// entry each; k_mesh's waves draw rays from the queue and search their meshes, k_finish finishes them, one dense lane per ray;
// MODE 2 ranks all rays of the tile (the order needs every ray's bin) and writes the sort keys.  Same arithmetic, same bits.
// FAST: the options that are run-time values in the general kernel are compile-time constants for the common case -- no
// textures, material sort on, candidate masks and all scene tables in LDS, no BVH mesh, no bump map, no depth
// of field, batched radiance buffers, not the cache-filling pass -- so that every test of them, and the code behind the
// untaken side, is gone (C4: k_bounce -6 %, the first bounce -11 %, fewer registers).  The host picks the variant per launch
// (enqueue_batch); everything else takes the general kernel, same results.  For the two halves of the split bounce (MODE 1, 2)
// FAST bakes only the subset that textured scenes with BVH meshes satisfy as well.
template <bool FIRST, int MODE, bool FAST = false>
__global__ __launch_bounds__(TILE, !FAST ? PT_BOUNCE_WAVES : MODE == 1 ? PT_FAST_WAVES_SPLIT : MODE == 2 ? PT_FAST_WAVES_SPLIT2 : FIRST ? PT_FAST_WAVES_FIRST : PT_FAST_WAVES) void k_bounce(const BounceParams p_in) {      // (the camera-ray
                                                                                   // variant needs 65 registers: seven waves without spilling)
#ifdef PT_WGCLOCK
    const unsigned long long wg_t0 = wall_clock64();       // 100 MHz: latency of the workgroup's phases (prologue, tile loop, tail)
#endif
    BounceParams p = p_in;
    if (FAST && MODE != 0) {             // the two halves of the split bounce: the subset that holds for textured BVH scenes too
        p.sort = 1; p.albedo = nullptr; p.emit_count = nullptr; p.sc.cull = 1; p.sc.tri_lds = 1;
    }
    if (FAST && MODE == 0) {
        p.uses_uv = PT_EXP_CARRY_UV; p.sort = 1; p.albedo = nullptr; p.emit_count = nullptr; p.dof = 0;
        p.sc.cull = 1; p.sc.tri_lds = 2; p.sc.bump_bits = 0; p.sc.ntri_lds = p.sc.ntri; p.sc.bvh_root = nullptr;
    }
    // dynamic LDS (pt_lds): [scene tables when staged: triangles, materials][2][WAVES][nbins] ranking histogram [2][nbins] running prefix
    // [nbins] tile counts [nbins+1] tile offsets [17][TILE] records being sorted
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int nb = p.nbins;
    const int triWords = (MODE != 2 && p.sc.tri_lds) ? sceneLdsWords(p.sc) : 0;      // (pass 2 of the split bounce only ranks: no scene tables)
    int32_t *lds = pt_lds + triWords;
    int32_t *w_all = lds, *w_scat = lds + WAVES * nb;              // (the ranking pass alternates between this histogram and a second one: below)
    int32_t *run_all = lds + 2 * WAVES * nb, *run_scat = run_all + nb;
    int32_t *tcs = run_scat + nb, *toff = tcs + nb;                 // stored-path count per bin of this tile, its prefix
    int32_t *tcnt = toff + nb + 1;                                  // tileIntersect's list counters [2][4], zero between uses
    int32_t *win = tcnt + 8;                                        // window over the input's run tables [WIN_WORDS] (windowLoad)
    int tq = 0;
    int32_t *rec = lds + ldsHeadWords(nb);                  // [17][TILE] record transpose buffer / tileIntersect scratch,
                                                                    // 16-byte aligned (64-bit LDS atomics live in it)
    constexpr int R_PIX = (FAST && MODE == 0 ? REC_ROWS_FAST0 : 17) - 3, R_MG = R_PIX + 1, R_KEY = R_PIX + 2;      // rows of the record buffer (12 floats, [u, v,] these)
    uint32_t *qbuf = reinterpret_cast<uint32_t *>(rec + REC_WORDS);   // MODE 1: LDS stage of the queue of parked rays, [QCAP], kept
    int32_t *qcnt = rec + REC_WORDS + QCAP, *qbase = qcnt + 1;        // across tiles (so not inside the record buffer)
    if (MODE == 1 && tid == 0) *qcnt = 0;
    for (int k = tid; k < 2 * nb; k += TILE) run_all[k] = 0;
    if (tid < 8) tcnt[tid] = 0;
    __syncthreads();
    const int seg = blockIdx.y;
    const int iter = p.iter + seg * p.iter_stride;
    const PathSoA in_k = soa_offset(p.in, p.seg_in * seg), stage_k = soa_offset(p.stage, p.seg_stage * seg);
    int32_t *counts_all = p.counts_all + p.seg_counts * seg, *counts_scat = p.counts_scat + p.seg_counts * seg;
    // stored paths per tile (a third block behind the two prefix tables): the tail reads only that many keys of a tile -- slots beyond hold
    // no record and, since round 5, no "no record" key either (4 B written and 4 B read per ended path of a 256-path tile saved)
    int32_t *tile_np = counts_all + 2 * (size_t)nb * p.maxTiles;
    // The tile epilogue without LDS (PT_DIRECT_STORE; up to 64 bins: lane b of EVERY wave owns bin b).  Until round 5 the stored paths of a
    // tile were written into an LDS record buffer at their slots, and read back in slot order for dense stores, behind wave 0's scan of
    // the per-bin counts: three barriers (counts ready, records in LDS, buffer free again) after the ranking's own, every one of them a
    // wait for the slowest of four waves -- 18 % of k_bounce's wave cycles sat in "ranking" and 8 % in "sort + write" for ~230 vector
    // instructions.  Now every wave sums the four waves' counts and scans them itself (the same ~20 instructions, nobody waits for wave
    // 0), a lane fetches its bin's offset from lane `bin` (ds_bpermute) and stores its record's quads straight to slot offset + rank:
    // lanes of one bin are consecutive slots, a wave's store is a handful of contiguous runs.  What is left per tile: the two barriers
    // of the pair test and the ranking's one.  The histogram rows a wave zeroes are now its OWN, after the pair test's barriers (every
    // wave has then left the previous tile): no other wave can still be reading them.
    const bool direct = PT_DIRECT_STORE && MODE != 2 && (FAST || nb <= 64);      // (the specialised variants are only launched with <= 64 bins: fast_violation)
    int32_t *chunk_out = p.chunk + p.seg_chunk * seg;
    int32_t *super_all = p.super_all + p.seg_totals * seg, *super_scat = p.super_scat + p.seg_totals * seg;
    int32_t *totals_all = p.totals_all + p.seg_totals * seg, *totals_scat = p.totals_scat + p.seg_totals * seg;
    float *part = (FAST || p.part) ? p.part + p.seg_part * seg : nullptr;
    const bool batched = FAST || part != nullptr;
    const int32_t *in_totals = FIRST ? nullptr : p.in_totals + p.seg_in_totals * seg;
    const int32_t *in_chunk = FIRST ? nullptr : p.in_chunk + p.seg_in_chunk * seg;
    const int in_nruns = nb * p.in_gx;
    // the input's stored paths per bin: lane b of every wave holds bin b's (ONE load instead of a chain of scalar ones -- a small launch's
    // workgroup lives 30 us, and this prologue is a third of it); more than 64 bins: the scalar loop
    int v_tot = 0;
    if (!FIRST && nb <= 64 && lane < nb) v_tot = in_totals[nb + lane];
    const int n_in = FIRST ? p.tm.owned : nb <= 64 ? __builtin_amdgcn_readlane(waveInclusiveScan(v_tot, lane), 63) : sum_totals(in_totals + nb, nb);
    const int ntiles = (n_in + TILE - 1) / TILE;
    // every workgroup owns a contiguous chunk of tiles, so that the prefix of a tile is (prefix of its chunk) +
    // (running sum inside the chunk) and no separate scan pass over the tiles is needed
    const int chunk = (ntiles + (int)gridDim.x - 1) / (int)gridDim.x;
    const int tile0 = min((int)blockIdx.x * chunk, ntiles), tile1 = min(tile0 + chunk, ntiles);
    // Which sorted positions hold records WITH an incoming direction (dir_bins: scatterRay reads it for reflective and refractive materials
    // and on OBJ geoms; a diffuse hit on a cube or sphere -- most of a Cornell scene -- never does, and its record's three direction
    // words are neither stored nor loaded: 24 B of the 120 a stored path moves).  The stream is sorted by bin: up to two ranges of
    // positions, from the input's per-bin totals.  In the same way the records of a material that only cubes have carry a 3-bit code of the
    // cube's tabulated normal (in pix's bits 28-30) instead of the normal: another 24 B; a diffuse wall's record is 32 B instead of 56.
    // (ptx_create leaves at most two runs of set bits in either mask, so two ranges always do: empty bins only merge them)
    auto binRanges = [&](unsigned long long mask, int &lo0, int &len0, int &lo1, int &len1) {
        int lo[2] = {0, 0}, hi[2] = {0, 0}, nr = 0, pos = 0;
        bool open = false;
        for (int b = 0; b < nb; b++) {
            const int tot = __builtin_amdgcn_readlane(v_tot, b);      // (masks are only in use with <= 64 bins)
            if (tot > 0) {
                const bool set = (mask >> b) & 1ull;
                if (set && !open) { if (nr < 2) lo[nr] = pos; open = true; }
                else if (!set && open) { if (nr < 2) hi[nr] = pos; nr++; open = false; }
            }
            pos += tot;
        }
        if (open) { if (nr < 2) hi[nr] = pos; nr++; }
        lo0 = lo[0]; len0 = hi[0] - lo[0]; lo1 = lo[1]; len1 = hi[1] - lo[1];
    };
    const bool masks_on = MODE != 2 && nb <= 64;
    const bool dir_some = masks_on && p.dir_bins != ~0ull;        // (uniform: the writer's side of the same rules.  Split bounce: the rays pass 1
    const bool ntab_some = masks_on && p.ntab_bins != 0ull;       // parks keep their direction -- k_mesh walks with it -- whatever their bin turns out to be)
    int dir_lo0 = 0, dir_len0 = 0x7fffffff, dir_lo1 = 0, dir_len1 = 0;      // sorted positions whose records carry a direction: all, unless ...
    int ntab_lo0 = 0, ntab_len0 = 0, ntab_lo1 = 0, ntab_len1 = 0;           // ... whose records carry a normal code instead of a normal: none, unless ...
    if (!FIRST && masks_on && p.in_dir_bins != ~0ull) binRanges(p.in_dir_bins, dir_lo0, dir_len0, dir_lo1, dir_len1);
    if (!FIRST && masks_on && p.in_ntab_bins != 0ull) binRanges(p.in_ntab_bins, ntab_lo0, ntab_len0, ntab_lo1, ntab_len1);
#ifdef PT_STAMPS
    unsigned long long st_acc[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, st_t0, st_t1;
#define STAMP(k) do { st_t1 = __builtin_amdgcn_s_memtime(); st_acc[k] += st_t1 - st_t0; st_t0 = st_t1; } while (0)
#else
#define STAMP(k) do { } while (0)
#endif
    // Head of the sort: the run that holds this workgroup's first sorted position, found by wave 0 in TWO memory round trips while
    // the other waves stage the scene tables (the short kernels of a small tile are chains of such round trips: the prologue was
    // a third of a workgroup's life there).  Trip 1: the flat [bin][group of 64 workgroups] table, eight entries per lane -> the
    // group.  Trip 2: that group's 64 runs and the 64 after them, three tables -> the run, and from the same registers the window
    // of the 64 runs from it on (what windowLoad would fetch in a third trip).
    if (!FIRST && MODE != 2 && tile0 < tile1) {
        if (wave == 0) {
            const int32_t *in_super = p.in_super + p.seg_in_totals * seg;
            const int A = tile0 * TILE;
            int pre_s = 0, pre_a = 0;
            // (rows of the group table are nsuper wide, entries past a bin's last group are zero: scanned as one flat array)
            int bs = scanFind8(in_super + (size_t)nb * p.nsuper, in_super, nb * p.nsuper, A, pre_s, pre_a, lane);      // A < n_in: there is one
            bs = bs < 0 ? 0 : bs;      // (cannot happen while the tables are what a k_bounce leaves; no address may depend on that)
            const int b = bs / p.nsuper, sg = bs - b * p.nsuper;
            const int rs = b * p.in_gx + sg * 64;                     // first run of the group; its 64 runs hold position A
            int cs[2], ca[2], cb[2];
#pragma unroll
            for (int h = 0; h < 2; h++) {
                const int r = rs + h * 64 + lane;
                // (a bin's last group may hold fewer than 64 runs: what follows in the flat table is the next bin's first runs, which
                // are also what follows in sorted order; position A itself lies in the group, i.e. in the first 64)
                const bool in = r < in_nruns;
                ca[h] = in ? in_chunk[r] : 0; cs[h] = in ? in_chunk[p.chunk_cap + r] : 0; cb[h] = in ? in_chunk[2 * p.chunk_cap + r] : 0;
            }
            int xs[2], xa[2];                                         // exclusive prefixes of the 128 entries
            const int is0 = waveInclusiveScan(cs[0], lane), ia0 = waveInclusiveScan(ca[0], lane);
            const int is1 = waveInclusiveScan(cs[1], lane), ia1 = waveInclusiveScan(ca[1], lane);
            const int t0s = __builtin_amdgcn_readlane(is0, 63), t0a = __builtin_amdgcn_readlane(ia0, 63);
            xs[0] = pre_s + is0 - cs[0]; xa[0] = pre_a + ia0 - ca[0];
            xs[1] = pre_s + t0s + is1 - cs[1]; xa[1] = pre_a + t0a + ia1 - ca[1];
            const unsigned long long m = __ballot(pre_s + is0 > A);
            const int w = m ? __ffsll((long long)m) - 1 : 0;          // the run, as an offset into the group
#pragma unroll
            for (int h = 0; h < 2; h++) {
                const int e = h * 64 + lane - w;                      // this entry's place in the window that starts at run rs + w
                if (e >= 0 && e < WIN) { win[e] = xs[h]; win[WIN + 1 + e] = xa[h]; win[2 * (WIN + 1) + e] = cb[h]; }
                if (e == WIN) { win[WIN] = xs[h]; win[2 * WIN + 1] = xa[h]; win[2 * (WIN + 1) + WIN] = rs + w + WIN; }
            }
        } else if (p.sc.tri_lds) stageSceneToLds(p.sc, tid - 64, TILE - 64);
        __syncthreads();
    } else if (MODE != 2 && tile0 < tile1 && p.sc.tri_lds) {      // (a workgroup without tiles shades nothing and needs no tables; nor does pass 2)
        stageSceneToLds(p.sc, tid, TILE);
        __syncthreads();
    }
#ifdef PT_WGCLOCK
    const unsigned long long wg_t1 = wall_clock64();
#endif
    struct InRec { float f[14]; int32_t pix, mg, idx; };
    // sorted position -> (entry of the local index, RNG stream index of its run's first survivor).  Uniform call: the window
    // moves on (barriers) when the tile's last position lies beyond it -- a few times per workgroup at most.
    auto locate = [&](int tile_, uint32_t &li4, int &idx_base) {
        const int jp = min(tile_ * TILE + tid, n_in - 1), last = min(tile_ * TILE + TILE - 1, n_in - 1);
        bool done = false;
        for (;;) {
            const int wend = win[WIN];
            if (!done && jp < wend) {
                int k = 0;
#pragma unroll
                for (int st = WIN / 2; st; st >>= 1) k += win[k + st] <= jp ? st : 0;      // last run that starts at or before jp
                // (the clamp cannot bite while the tables are consistent: it is there so that no gather address depends on that)
                const uint32_t li = (uint32_t)(win[2 * (WIN + 1) + k] + (jp - win[k]));
                if (__builtin_expect(li >= p.fence_slots, 0)) fence_report(p);
                li4 = min(li, p.fence_slots - 1u) << 2;
                idx_base = win[WIN + 1 + k];
                done = true;
            }
            // (uniform: every thread reads the same window; the second test cannot hold while the tables are what a k_bounce
            // leaves -- position < n_in lies in some run -- it is there so that the loop ends whatever they hold)
            if (last < wend || win[2 * (WIN + 1) + WIN] >= in_nruns) break;
            __syncthreads();
            if (wave == 0) windowLoad(win, in_chunk, p.chunk_cap, in_nruns, win[2 * (WIN + 1) + WIN], wend, win[2 * WIN + 1], lane);
            __syncthreads();
        }
    };
    auto fetch = [&](uint32_t li4, int idx_base, InRec &r, int jp) -> bool {
        const PathSoA in = soa_fresh(in_k);
        // the sorted stream is not materialised: its position is entry li of the previous bounce's local index, which names the
        // slot of that bounce's stage and the path's rank inside its run
        const uint32_t w = (uint32_t)ld_u(in.lsrc(), li4);
        uint32_t j;
        if (p.in_idx16) { j = (li4 >> 2) + (uint32_t)(int32_t)(int16_t)(w & 0xffffu); r.idx = idx_base + (int)(w >> 16); }
        else { j = w; r.idx = idx_base + ld_u(in.lidx(), li4); }
        if (__builtin_expect(j >= p.fence_slots, 0)) fence_report(p);
        const uint32_t j4 = min(j, p.fence_slots - 1u) << 2;
        const bool with_dir = (uint32_t)(jp - dir_lo0) < (uint32_t)dir_len0 || (uint32_t)(jp - dir_lo1) < (uint32_t)dir_len1;
        const bool coded_n = (uint32_t)(jp - ntab_lo0) < (uint32_t)ntab_len0 || (uint32_t)(jp - ntab_lo1) < (uint32_t)ntab_len1;
        typedef float quad __attribute__((ext_vector_type(4)));
        const uint32_t j16 = j4 << 2;
        const quad A = ld_u(reinterpret_cast<const quad *>(in.quadA()), j16), B = ld_u(reinterpret_cast<const quad *>(in.quadB()), j16);
        r.f[0] = A.x; r.f[1] = A.y; r.f[2] = A.z; r.pix = __float_as_int(A.w);
        r.f[6] = B.x; r.f[7] = B.y; r.f[8] = B.z; r.mg = __float_as_int(B.w);
        r.f[3] = r.f[4] = r.f[5] = 0.f; r.f[9] = r.f[10] = r.f[11] = 0.f; r.f[12] = r.f[13] = 0.f;
        // (texcoords matter on OBJ geoms only, whose records have both a normal and a direction: where a part is not read they are the 0 it would hold)
        if (with_dir) { const quad D = ld_u(reinterpret_cast<const quad *>(in.quadD()), j16); r.f[3] = D.x; r.f[4] = D.y; r.f[5] = D.z; if (p.uses_uv) r.f[13] = D.w; }
        if (!coded_n) { const quad C = ld_u(reinterpret_cast<const quad *>(in.quadC()), j16); r.f[9] = C.x; r.f[10] = C.y; r.f[11] = C.z; if (p.uses_uv) r.f[12] = C.w; }
        if (coded_n) {      // the cube's tabulated normal, the words decodeKey took it from (the tile path: the tables are staged)
            // (the geom index comes out of the same untrusted record as the pixel slot below: clamped before it indexes the tables)
            const int g = (int)min((uint32_t)(r.mg >> 16), (uint32_t)(p.sc.ngeoms - 1));
            const vec3 n = cubeNormalByCode(p.sc, reinterpret_cast<const float *>(pt_lds) + p.sc.ntri_lds * 24 + p.sc.nmats * 11, g, (r.pix >> 28) & 7);
            r.f[9] = n.x; r.f[10] = n.y; r.f[11] = n.z;
            r.pix &= 0x0fffffff;
        }
        // (fence: the pixel slot becomes the address of the path's radiance when it ends -- found by a record read with other masks than
        // it was written with, end of round 4: a normal code taken for part of the slot.  A fenced record is a DEAD path, as in k_finish:
        // it is counted, and it neither scatters nor adds light to a pixel that is not its own)
        if (__builtin_expect((uint32_t)r.pix >= (uint32_t)p.tm.owned, 0)) { fence_report(p); r.pix = 0; return false; }
        return true;
    };
    auto classifyRay = [&](const Hit &hit, const PathState &ps, int pix, int &bin, bool &pending) {
        classifyPath<FIRST>(p, iter, part, batched, hit, ps.color, pix, bin, pending);
    };
    int32_t *ccnt = qcnt + 2;                                   // MODE 1: candidates of the tile so far (LDS)
    int32_t k1_next = 0;                                        // MODE 2: the next tile's word, requested one tile ahead
    // The ranking pass (MODE 2) with ONE barrier per tile (round 5; three until then: counts by wave 0, keys scattered to their slots through
    // LDS).  What it owes the tail is a key per STORED path at the path's slot; the slot is in the word pass 1 / k_finish left, so the key
    // is stored there directly.  Slots without a record need no "-1" any more: the tail reads a tile's keys only where records can lie --
    // pass 1's at the bottom, the parked rays' at the top (tile_np: both counts) -- and k_finish marks the parked rays that ended.  The
    // per-bin counts are bookkeeping of their owner thread (thread b owns bin b, every tile), nobody waits for them.  The histogram
    // alternates between two buffers, each wave zeroing its OWN rows at the start of a tile: whoever still sums the previous tile's
    // reads the other buffer, and the tile before that lies behind the previous tile's barrier.
    // MEASURED AND NOT KEPT (PT_RANK_ONE_BARRIER 0; tools/runs/r5f.sh, one box, Lit = three barriers / Rank1 = this form): the pass
    // alone 0.147 -> 0.140 ms per iteration of C5, the wall 0.893 -> 0.904.  The pass is not a chain of barriers after all -- at 4K it
    // moves 0.45 GB per launch in 0.2 ms -- and keys scattered 4 B at a time cost the three launch sets more than its barriers did.
    constexpr bool RANK1 = MODE == 2 && PT_RANK_ONE_BARRIER;
    int rank_par = 0;
    if (MODE == 2 && !RANK1) {                                  // (its histogram: zeroed here once, then after every tile's ranking)
        for (int k = tid; k < 2 * WAVES * nb; k += TILE) lds[k] = 0;
        __syncthreads();
    }
    if (MODE == 2 && !FIRST && tile0 < tile1 && tile0 * TILE + tid < n_in) k1_next = ld_u(soa_fresh(stage_k).lsrc(), (uint32_t)(tile0 * TILE + tid) << 2);
    for (int tile = tile0; tile < tile1; tile++) {
#ifdef PT_STAMPS
        st_t0 = __builtin_amdgcn_s_memtime();
#endif
        const int i = tile * TILE + tid;
        bool alive = i < n_in;
        // camera rays: the geoms this tile's 256 pixels can see at all (host, update_tile_geoms: conservative screen rectangles of the
        // geoms' world boxes): the per-ray candidate masks test only these -- most tiles see two or three of a Cornell scene's seven
        uint32_t tile_subset = 0xffffffffu;
        if (FIRST && MODE != 2 && p.tile_geoms) tile_subset = ((const __attribute__((address_space(4))) uint32_t *)p.tile_geoms)[tile];
        InRec cur;
        uint32_t li4 = 0;
        int idx_base = 0;
        if (!FIRST && MODE != 2) locate(tile, li4, idx_base);
        const PathSoA stage = soa_fresh(stage_k);      // field addresses are formed where they are used
        PathState ps;
        int pix = 0;                 // slot among the owned pixels: what the path carries instead of the pixel index
        // ranking histogram (read after later barriers).  The ranking pass (MODE 2) has no intersection whose barriers would separate
        // this from the ballots' writes: it zeroes the histogram right after a tile's LAST read of it instead (below), and once before
        // its first tile -- three barriers per tile instead of five for a kernel that is a chain of barriers and little else.
        if (MODE != 2 && !direct) for (int k = tid; k < 2 * WAVES * nb; k += TILE) lds[k] = 0;
        ps.o = ps.d = ps.color = V3(0.f, 0.f, 0.f);
        unsigned long long key = KEY_NONE;
        int32_t k1 = 0;
        if (MODE == 2 && !FIRST) {
            // the word of the NEXT tile is requested before this tile's is used: the ranking pass is otherwise one exposed memory
            // round trip per tile (pass 2 at 4K: 110 -> 85 us per launch; not on the first bounce, most of whose tiles pass 1 finished)
            k1 = k1_next;
            if (tile + 1 < tile1 && i + TILE < n_in) k1_next = ld_u(soa_fresh(stage_k).lsrc(), (uint32_t)(i + TILE) << 2);
        }
        if (MODE == 2 && FIRST && p.tile_done && (p.tile_done + (size_t)p.maxTiles * seg)[tile]) {
            // pass 1 finished this tile (records and keys are in the stage): only its per-bin counts, which pass 1 left in
            // the prefix tables, are folded into this workgroup's running prefix
            for (int b = tid; b < nb; b += TILE) {
                const int ca = counts_all[(size_t)b * p.maxTiles + tile], cs = counts_scat[(size_t)b * p.maxTiles + tile];
                counts_all[(size_t)b * p.maxTiles + tile] = run_all[b];
                counts_scat[(size_t)b * p.maxTiles + tile] = run_scat[b];
                run_all[b] += ca;
                run_scat[b] += cs;
            }
            if (!RANK1) __syncthreads();      // (thread b is bin b's owner in every tile: nobody else reads the running prefixes before the tail)
            continue;
        }
        if (RANK1) {                          // this tile's histogram buffer, this wave's rows of it
            w_all = rank_par ? rec : lds; w_scat = w_all + WAVES * nb;
            rank_par ^= 1;
            for (int k = lane; k < nb; k += 64) { w_all[wave * nb + k] = 0; w_scat[wave * nb + k] = 0; }
        }
        int bin = -1;
        bool pending = false, pass1_partial = false;
        int myslot = 0;                  // split bounce: the slot (inside the tile) of this ray's parked state / stored record
        if (MODE == 1 && tid == 0) *ccnt = 0;           // (first touched after tileIntersect's barriers)
        if (MODE == 2) {                 // pass 1 left one word per ray; only the rays with mesh candidates were parked
            alive = false;
            // (every ray is finished by now: by pass 1, or -- the ones with mesh candidates -- by k_finish)
            if (i < n_in) {
                if (FIRST) k1 = ld_u(stage.lsrc(), (uint32_t)i << 2);
                myslot = (k1 >> 16) & (TILE - 1);
                alive = (k1 & K1_ALIVE) != 0; pending = (k1 & K1_PEND) != 0; bin = k1 & 0xffff;
            }
        } else if (alive) {
            if (FIRST) {
                int x, y;
                owned_pixel(p.tm, i, x, y);
                pix = i;                 // the slot; the pixel is (x, y)
                // (a tile that sees no geom at all: its rays miss whatever they are -- none is generated, none is tested)
                if (tile_subset != 0u) generateRay(p.cam, iter, p.traceDepth, p.aa != 0, p.dof != 0, x, y, ps);
            } else {
                // shadeFakeMaterial for a path that is known to scatter (src/pathtrace.cu:391-394)
                const bool rec_ok = fetch(li4, idx_base, cur, min(i, n_in - 1));
                const vec3 intersect = V3(cur.f[0], cur.f[1], cur.f[2]);           // stored as origin + t * direction
                ps.d = V3(cur.f[3], cur.f[4], cur.f[5]);
                ps.color = V3(cur.f[6], cur.f[7], cur.f[8]);
                pix = cur.pix;
                Hit h;
                h.t = 1.f;
                h.n = V3(cur.f[9], cur.f[10], cur.f[11]);
                h.u = cur.f[12]; h.v = cur.f[13];
                const int mg = cur.mg;
                h.mat = mg & 0xffff; h.geom = mg >> 16;
                const int sidx = cur.idx;
#ifdef PT_STAMPS
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                STAMP(11);
#endif
                Rng rng; rng.seed(iter, sidx, 0);
                bool ended = (FAST && MODE == 0) ? scatterRay<false>(p.sc, ps, intersect, h, getMaterial(p.sc, h.mat), rng)
                                                 : scatterRay<true>(p.sc, ps, intersect, h, getMaterial(p.sc, h.mat), rng);
                if (ended && rec_ok) deposit(p.tm, p.image, part, batched, pix, ps.color, p.apps, p.fence_slots_cap);      // emissive texel: remainingBounces 1 -> 0, colour goes to the image
                if (ended || !rec_ok) alive = false;
            }
        }
        STAMP(0);        // load + shade (or ray generation)
        // computeIntersections(b) + the terminal cases of shadeFakeMaterial(b)
        Hit hit;
        hit.t = -1.f; hit.n = V3(0.f, 0.f, 0.f); hit.u = hit.v = 0.f; hit.geom = 0; hit.mat = 0; hit.ncode = 0;
        if (FIRST && MODE == 0 && tile_subset == 0u) {
            // Camera rays of a tile into which no geom's box projects (the wide margins of the Cornell frames: a third of C4's tiles):
            // every ray misses.  Nothing is generated, tested, ranked or stored -- the paths end black (their slot of the radiance buffer is
            // written), the tile's keys say "no record", and what moves on is the count of survivors in the miss bin (material 0), which
            // the stream indices of the next bounce and the ray statistics are made of.  (The per-tile prefix tables are only ever read
            // for stored paths: this tile has none.)
            int mbin = 0;
            bool mpend = false;
            if (alive) classifyRay(hit, ps, pix, mbin, mpend);
            if (tid == 0) tile_np[tile] = 0;
            if (tid == 0) run_all[p.sort ? p.sc.nmats - 1 : 0] += min(TILE, n_in - tile * TILE);
            continue;
        }
        if (FIRST && MODE == 1 && tile_subset == 0u && p.tile_done) {
            // the same in pass 1 of the split bounce: the tile is "finished in pass 1" (tile_done) with nothing stored; its per-bin counts --
            // all survivors in the miss bin -- go into the prefix tables as they are, pass 2 folds them into its running prefix
            int mbin = 0;
            bool mpend = false;
            if (alive) classifyRay(hit, ps, pix, mbin, mpend);
            if (tid == 0) tile_np[tile] = 0;
            const int missbin = p.sort ? p.sc.nmats - 1 : 0, nalive = min(TILE, n_in - tile * TILE);
            for (int b = tid; b < nb; b += TILE) {
                counts_all[(size_t)b * p.maxTiles + tile] = b == missbin ? nalive : 0;
                counts_scat[(size_t)b * p.maxTiles + tile] = 0;
            }
            if (tid == 0) (p.tile_done + (size_t)p.maxTiles * seg)[tile] = 1;
            continue;
        }
        {
            Ray ray; ray.o = ps.o; ray.d = ps.d;
            uint32_t mesh_cand = 0;
            if (MODE == 1) {
                if (FIRST && tile_subset == 0u) __syncthreads();      // (the histogram is zeroed: what tileIntersect's barriers see to otherwise)
                else tileIntersect<true, false, FIRST>(p.sc, alive, ray, p.uses_uv != 0, hit, rec, tcnt, tq, tid, lane, wave, key, mesh_cand TI_PASS, tile_subset);
                // Camera rays are coherent: most tiles of the first bounce (256 neighbouring pixels of a row) hold no ray that
                // reaches a mesh's box at all.  Such a tile is finished right here -- winner's normal, terminal cases,
                // ranking, in-tile sort, stage write, as in the unsplit kernel -- instead of being parked and picked up again;
                // its per-bin counts go into the prefix tables as they are, pass 2 folds them into its running prefix.
                bool finish_here = false;
                if (FIRST && p.tile_done) {
                    finish_here = !__syncthreads_or(mesh_cand != 0u);
                    if (tid == 0) (p.tile_done + (size_t)p.maxTiles * seg)[tile] = finish_here ? 1 : 0;
                }
                if (finish_here) {
                    if (alive) decodeKey<true>(p.sc, reinterpret_cast<const float *>(pt_lds) + p.sc.ntri_lds * 24 + p.sc.nmats * 11, key, ray,
                                         p.uses_uv != 0, hit);
                    goto classify;
                }
                // A ray WITH mesh candidates is parked -- origin, direction, colour, pixel, best key so far, in a slot of its own
                // counted down from the top of the tile -- and its candidates are queued with that slot.  A ray without -- 93 % of
                // them past the first bounce -- is finished right here like the unsplit kernel would: nearest hit, terminal cases,
                // ranked and sorted by bin among its like, record written to the bottom of the tile; pass 2 gets ONE word about it
                // (round 3; rounds 1-2 parked every ray: 48 B out and 48 B back in for each, which is what bounded the two passes).
                // What pass 2 still does for all is the ranking that defines the order: it needs every ray's bin, and the
                // candidates' are not known before k_mesh / k_finish.
                pass1_partial = true;
                {
                    const bool is_cand = mesh_cand != 0u;                        // (implies alive)
                    const unsigned long long cb = __ballot(is_cand);
                    int cbase = 0;
                    if (lane == 0 && cb) cbase = atomicAdd(ccnt, __popcll(cb));
                    myslot = TILE - 1 - (__builtin_amdgcn_readfirstlane(cbase) + wavePrefix(cb, lane));
                    const int sa = tile * TILE + myslot;
                    if (is_cand) {
                        stage.px()[sa] = ray.o.x; stage.py()[sa] = ray.o.y; stage.pz()[sa] = ray.o.z;
                        stage.dx()[sa] = ray.d.x; stage.dy()[sa] = ray.d.y; stage.dz()[sa] = ray.d.z;
                        stage.cr()[sa] = ps.color.x; stage.cg()[sa] = ps.color.y; stage.cb()[sa] = ps.color.z;
                        stage.pix()[sa] = pix;
                        stage.mg()[sa] = i;                                      // whose ray this is: k_finish writes the verdict to lsrc[i]
                        stage.nx()[sa] = __int_as_float((int)mesh_cand);         // the meshes whose boxes it reaches (bit per geom)
                        (p.keys + p.seg_keys * seg)[sa] = key;
                        k1 = K1_CAND | (myslot << 16);
                    }
                    // one queue entry per parked ray -- its slot; which meshes it is a candidate for travels with the ray -- through an
                    // LDS buffer (behind the record buffer) to the global queue in blocks: one global atomic per ~50 tiles instead of
                    // one per wave (a single hot counter)
                    int base = 0;
                    if (lane == 0 && cb) base = atomicAdd(qcnt, __popcll(cb));
                    const int qi = __builtin_amdgcn_readfirstlane(base) + wavePrefix(cb, lane);      // (lane 0's value: read by all lanes)
                    if (is_cand) qbuf[qi] = (uint32_t)sa;
                    if (is_cand) alive = false;                                  // not part of pass 1's ranking and records
                }
                if (alive) decodeKey<true>(p.sc, reinterpret_cast<const float *>(pt_lds) + p.sc.ntri_lds * 24 + p.sc.nmats * 11, key, ray,
                                     p.uses_uv != 0, hit);
                goto classify;
            } else if (MODE == 2) {
                if (!RANK1) rec[tid] = -1;                        // this slot's key, until a stored path claims the slot (keybuf below;
                                                                  // the barriers of the ranking lie between this and the claims)
            } else if (p.sc.cull) {
                // The specialised kernel is compiled for PT_FAST_WAVES waves per SIMD, i.e. 72 registers.  The thread's own state
                // that is only needed again after the intersection -- throughput colour and pixel slot; the ray itself is in
                // tileIntersect's LDS copy anyway -- therefore waits in a free part of the record buffer instead of in
                // registers: the pair tests are where the register demand peaks.  (MODE 1 needs more than parking frees and
                // stays at 4 waves, where parking only costs LDS traffic; MODE 2 fits 96 registers as it is.)
                constexpr bool PARK = FAST && PT_PARK_STATE;
                float *park = reinterpret_cast<float *>(rec) + 12 * TILE;
                {
                    if (PARK) {
                        park[0 * TILE + tid] = ps.color.x; park[1 * TILE + tid] = ps.color.y; park[2 * TILE + tid] = ps.color.z;
                        rec[15 * TILE + tid] = pix;
                    }
                    tileIntersect<false, PARK, FIRST>(p.sc, alive, ray, p.uses_uv != 0, hit, rec, tcnt, tq, tid, lane, wave, key, mesh_cand TI_PASS, tile_subset);
                    if (PARK) {
                        asm volatile("" ::: "memory");
                        const float *rb = reinterpret_cast<const float *>(rec);
                        ps.o = V3(rb[0 * TILE + tid], rb[1 * TILE + tid], rb[2 * TILE + tid]);
                        ps.d = V3(rb[3 * TILE + tid], rb[4 * TILE + tid], rb[5 * TILE + tid]);
                        ps.color = V3(park[0 * TILE + tid], park[1 * TILE + tid], park[2 * TILE + tid]);
                        pix = rec[15 * TILE + tid];
                    }
                }
            }
            else {
                if (alive) intersectScene(p.sc, ray, hit);
                __syncthreads();                                  // histogram zeroed (tileIntersect has barriers of its own)
            }
        }
        STAMP(1);        // intersect
    classify:
        // (direct epilogue: this wave's rows of the ranking histogram -- every path to here has passed a barrier of THIS tile, so no wave
        // is still summing the previous tile's)
        if (MODE != 2 && direct && lane < nb) { w_all[wave * nb + lane] = 0; w_scat[wave * nb + lane] = 0; }
        if (MODE != 2 && alive) classifyRay(hit, ps, pix, bin, pending);
        STAMP(2);        // classify + deposit
        // stable rank of this path inside its tile, per material bin: among all alive paths (-> RNG stream
        // index) and among the stored ones (-> storage position)
        int r_all = 0, r_scat = 0;
        {
            // a lane's bin as ONE integer per question (-1: not part of it): a bin's lanes are then a single v_cmp_eq into a scalar pair
            // (a ballot of `flag && bin == b` makes the compiler materialise the flag first), and the rank among them two v_mbcnt: 15
            // vector instructions per bin that occurs in the wave instead of 28.  (Visiting EVERY bin in turn instead -- no readlane,
            // no find-first -- is 13 per bin and loses: camera rays see two or three of the eight bins.)
            const int abin = alive ? bin : -1, pbin = pending ? bin : -1;
            if (PT_RANK_SLICED && (MODE != 2 || RANK1) && !FIRST && nb <= 16) {
                // Up to 16 bins, later bounces (a wave of scattered rays sees four or five of the bins): bit-sliced.  Four ballots give the
                // lanes whose bin has bit k set; a lane ANDs together, per bit of its OWN bin, that mask or its complement -- the lanes of
                // its bin, as a 64-bit value of its own -- and ranks itself with v_mbcnt on it: ~30 vector instructions whatever the number
                // of bins, where the loop below costs ~15 per bin that occurs.  (Camera rays see two or three bins: they keep the loop.)
                const unsigned long long A = __builtin_amdgcn_uicmp((uint32_t)abin, 0xffffffffu, 33), P = __builtin_amdgcn_uicmp((uint32_t)pbin, 0xffffffffu, 33);
                uint32_t slo = 0xffffffffu, shi = 0xffffffffu;
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    if (k && (nb - 1) >> k == 0) break;                       // (uniform: bins below 2^k need no more bits)
                    const unsigned long long Bk = __builtin_amdgcn_uicmp((uint32_t)bin & (1u << k), 0u, 33);
                    const bool set = (bin >> k) & 1;
                    slo &= set ? (uint32_t)Bk : ~(uint32_t)Bk;
                    shi &= set ? (uint32_t)(Bk >> 32) : ~(uint32_t)(Bk >> 32);
                }
                const uint32_t alo = slo & (uint32_t)A, ahi = shi & (uint32_t)(A >> 32), plo = slo & (uint32_t)P, phi = shi & (uint32_t)(P >> 32);
                const int ra = (int)__builtin_amdgcn_mbcnt_hi(ahi, __builtin_amdgcn_mbcnt_lo(alo, 0u));
                const int rs = (int)__builtin_amdgcn_mbcnt_hi(phi, __builtin_amdgcn_mbcnt_lo(plo, 0u));
                if (alive) {
                    r_all = ra; r_scat = rs;
                    if (ra == 0) {                                              // the first lane of its bin in this wave: the wave's counts
                        w_all[wave * nb + bin] = __popc(alo) + __popc(ahi);
                        w_scat[wave * nb + bin] = __popc(plo) + __popc(phi);
                    }
                }
            } else {
                unsigned long long remaining = __builtin_amdgcn_uicmp((uint32_t)abin, 0xffffffffu, 33);       // (alive lanes)
                while (remaining) {
                    int leader = __ffsll((long long)remaining) - 1;
                    int b = __builtin_amdgcn_readlane(abin, leader);      // (leader is wave-uniform: no LDS round trip as __shfl would make)
                    const unsigned long long m_all = __builtin_amdgcn_uicmp((uint32_t)abin, (uint32_t)b, 32), m_scat = __builtin_amdgcn_uicmp((uint32_t)pbin, (uint32_t)b, 32);
                    const int ra = wavePrefix(m_all, lane), rs = wavePrefix(m_scat, lane);
                    if (abin == b) { r_all = ra; r_scat = rs; }
                    if (lane == leader) {
                        w_all[wave * nb + b] = __popcll(m_all);
                        w_scat[wave * nb + b] = __popcll(m_scat);
                    }
                    remaining &= ~m_all;
                }
            }
        }
        STAMP(12);       // (ranking: ballots)
        __syncthreads();
        STAMP(13);       // (ranking: wait at its first barrier)
        if (alive) {
            for (int w = 0; w < wave; w++) { r_all += w_all[w * nb + bin]; r_scat += w_scat[w * nb + bin]; }
        }
        if (RANK1) {
            for (int b = tid; b < nb; b += TILE) {               // bin b's owner: the tile's place in the chunk
                int ca = 0, cs = 0;
                for (int w = 0; w < WAVES; w++) { ca += w_all[w * nb + b]; cs += w_scat[w * nb + b]; }
                counts_all[(size_t)b * p.maxTiles + tile] = run_all[b];
                counts_scat[(size_t)b * p.maxTiles + tile] = run_scat[b];
                run_all[b] += ca;
                run_scat[b] += cs;
            }
            if (pending) st_u(soa_fresh(stage_k).idx(), (uint32_t)(tile * TILE + myslot) << 2, stage_key(bin, r_all, r_scat));
            STAMP(3);
            continue;
        }
        int toff_bin = 0, npend_r = 0;       // direct epilogue: this path's bin's offset in the tile, the tile's stored paths
        if (direct) {
            int ln = lane;                   // (opaque, see below)
            asm volatile("" : "+v"(ln));
            int ca = 0, cs = 0;
            if (ln < nb) for (int w = 0; w < WAVES; w++) { ca += w_all[w * nb + ln]; cs += w_scat[w * nb + ln]; }
            if (wave == 0 && ln < nb) {      // the tile's place in the chunk: one wave's business, as before
                counts_all[(size_t)ln * p.maxTiles + tile] = MODE == 1 ? ca : run_all[ln];
                counts_scat[(size_t)ln * p.maxTiles + tile] = MODE == 1 ? cs : run_scat[ln];
                run_all[ln] += ca;
                run_scat[ln] += cs;
            }
            const int inc = waveInclusiveScan(cs, ln);
            npend_r = __builtin_amdgcn_readlane(inc, 63);              // (lanes >= nb add 0)
            toff_bin = __builtin_amdgcn_ds_bpermute((bin < 0 ? 0 : bin) << 2, inc - cs);
            STAMP(14);
        } else if (nb <= 64) {
            // wave 0: lane b owns bin b -- tile counts, running prefixes and the in-tile offsets by a wave scan
            if (wave == 0) {
                // `ln` = lane, but opaque to the optimiser: otherwise the per-lane addresses below are loop invariants,
                // get hoisted out of the tile loop, and -- the kernel being at its register limit -- are spilled to
                // scratch, whose reloads (memory latency, one after the other) then sit on every tile's critical path
                int ln = lane;
                asm volatile("" : "+v"(ln));
                int ca = 0, cs = 0;
                if (ln < nb) {
                    for (int w = 0; w < WAVES; w++) { ca += w_all[w * nb + ln]; cs += w_scat[w * nb + ln]; }
                    // (MODE 1 finishing a tile: the tile's own counts, for pass 2 to fold in)
                    counts_all[(size_t)ln * p.maxTiles + tile] = MODE == 1 ? ca : run_all[ln];
                    counts_scat[(size_t)ln * p.maxTiles + tile] = MODE == 1 ? cs : run_scat[ln];
                    run_all[ln] += ca;
                    run_scat[ln] += cs;
                }
                const int inc = waveInclusiveScan(cs, ln);
                if (ln < nb) toff[ln] = inc - cs;
                if (ln == nb - 1) toff[nb] = inc;
            }
            STAMP(14);   // (ranking: counts, wave 0's scan)
            __syncthreads();
        } else {
            for (int b = tid; b < nb; b += TILE) {
                int ca = 0, cs = 0;
                for (int w = 0; w < WAVES; w++) { ca += w_all[w * nb + b]; cs += w_scat[w * nb + b]; }
                counts_all[(size_t)b * p.maxTiles + tile] = MODE == 1 ? ca : run_all[b];
                counts_scat[(size_t)b * p.maxTiles + tile] = MODE == 1 ? cs : run_scat[b];
                run_all[b] += ca;
                run_scat[b] += cs;
                tcs[b] = cs;
            }
            __syncthreads();
            if (tid == 0) {
                int o = 0;
                for (int b = 0; b < nb; b++) { toff[b] = o; o += tcs[b]; }
                toff[nb] = o;
            }
            __syncthreads();
        }
        STAMP(3);        // ranking + counts
        if (MODE == 2) {                 // the records lie in their slots already (pass 1, or above): only the keys are left,
            int32_t *keybuf = rec;       // one per SLOT (through LDS: a path's slot is not its thread), -1 where no record lies
            for (int k = tid; k < 2 * WAVES * nb; k += TILE) lds[k] = 0;      // (w_all / w_scat were last read before the barriers above;
                                                                               // the barrier below is in front of the next tile's ballots)
            if (pending) keybuf[myslot] = stage_key(bin, r_all, r_scat);
            __syncthreads();
            // (stored paths fill the slots from the bottom -- pass 1's records and k_finish's parked ones are not contiguous: keys for every slot)
            st_u(soa_fresh(stage_k).idx(), (uint32_t)i << 2, keybuf[tid]);
            if (FIRST && tid == 0) tile_np[tile] = TILE;          // (this form writes a key for every slot; the camera bounce's tail looks the counts
                                                                  // up -- the tiles pass 1 finished have fewer keys --, a later bounce's knows)
            continue;
        }
        if (direct) {
            const PathSoA stage = soa_fresh(stage_k);
            const bool partial = MODE == 1 && pass1_partial;         // (pass 2 writes such a tile's keys and its count)
            if (partial && i < n_in) {
                if (alive) k1 = K1_ALIVE | bin | (pending ? K1_PEND | ((toff_bin + r_scat) << 16) : 0);
                st_u(stage.lsrc(), (uint32_t)i << 2, k1);
            }
            if (pending) {
                const uint32_t g4 = (uint32_t)(tile * TILE + toff_bin + r_scat) << 2, g16 = g4 << 2;
                const vec3 sp = add(ps.o, scale(ps.d, hit.t));      // the point shadeFakeMaterial will shade (:392)
                const bool with_dir = !dir_some || ((p.dir_bins >> bin) & 1ull);
                const bool coded_n = ntab_some && ((p.ntab_bins >> bin) & 1ull);
                typedef float quad __attribute__((ext_vector_type(4)));
                quad A, B;
                A.x = sp.x; A.y = sp.y; A.z = sp.z; A.w = __int_as_float(coded_n ? (pix | (hit.ncode << 28)) : pix);
                B.x = ps.color.x; B.y = ps.color.y; B.z = ps.color.z; B.w = __int_as_float(hit.mat | (hit.geom << 16));
                st_u(reinterpret_cast<quad *>(stage.quadA()), g16, A);
                st_u(reinterpret_cast<quad *>(stage.quadB()), g16, B);
                if (with_dir) {
                    quad D;
                    D.x = ps.d.x; D.y = ps.d.y; D.z = ps.d.z; D.w = p.uses_uv ? hit.v : 0.f;
                    st_u(reinterpret_cast<quad *>(stage.quadD()), g16, D);
                }
                if (!coded_n) {
                    quad C;
                    C.x = hit.n.x; C.y = hit.n.y; C.z = hit.n.z; C.w = p.uses_uv ? hit.u : 0.f;
                    st_u(reinterpret_cast<quad *>(stage.quadC()), g16, C);
                }
                if (!partial) st_u(stage.idx(), g4, stage_key(bin, r_all, r_scat));
            }
            // stored paths of the tile: pass 1's own lie in slots [0, n); a partial tile's parked rays in the top *ccnt slots (the ranking pass
            // writes the keys of those that go on, k_finish marks the ones that ended)
            if (tid == 0) tile_np[tile] = npend_r | (partial ? *ccnt << 16 : 0);
            STAMP(4);
            if (MODE == 1 && *qcnt > QCAP - TILE) flushQueue(p, seg, qbuf, qcnt, qbase, tid);      // (uniform: the tile's last atomic on it lies before the ranking's barrier)
            continue;
        }
        if (MODE == 1 && pass1_partial && i < n_in) {
            if (alive) k1 = K1_ALIVE | bin | (pending ? K1_PEND | ((toff[bin] + r_scat) << 16) : 0);
            st_u(soa_fresh(stage_k).lsrc(), (uint32_t)i << 2, k1);
        }
        // Stored paths go to the stage sorted by bin inside the tile (through LDS), so that both this write and
        // the tail's read are dense and coalesced and the next bounce's gather reads per-bin runs.
        if (pending) {
            const int slot = toff[bin] + r_scat;
            const vec3 sp = add(ps.o, scale(ps.d, hit.t));      // the point shadeFakeMaterial will shade (:392)
            float *rf = reinterpret_cast<float *>(rec);
            rf[0 * TILE + slot] = sp.x; rf[1 * TILE + slot] = sp.y; rf[2 * TILE + slot] = sp.z;
            if (!dir_some || ((p.dir_bins >> bin) & 1ull)) { rf[3 * TILE + slot] = ps.d.x; rf[4 * TILE + slot] = ps.d.y; rf[5 * TILE + slot] = ps.d.z; }
            rf[6 * TILE + slot] = ps.color.x; rf[7 * TILE + slot] = ps.color.y; rf[8 * TILE + slot] = ps.color.z;
            const bool coded_n = ntab_some && ((p.ntab_bins >> bin) & 1ull);
            if (!coded_n) { rf[9 * TILE + slot] = hit.n.x; rf[10 * TILE + slot] = hit.n.y; rf[11 * TILE + slot] = hit.n.z; }
            if (p.uses_uv) { rf[12 * TILE + slot] = hit.u; rf[13 * TILE + slot] = hit.v; }
            rec[R_PIX * TILE + slot] = coded_n ? (pix | (hit.ncode << 28)) : pix;
            rec[R_MG * TILE + slot] = hit.mat | (hit.geom << 16);
            rec[R_KEY * TILE + slot] = stage_key(bin, r_all, r_scat);
        }
        __syncthreads();
        {
            const PathSoA stage = soa_fresh(stage_k);
            const int npend = toff[nb];
            const uint32_t gi4 = (uint32_t)(tile * TILE + tid) << 2;
            if (tid < npend) {
                const float *rf = reinterpret_cast<const float *>(rec);
                const int32_t skey = rec[R_KEY * TILE + tid];
                // (this slot's record carries a direction iff its bin says so: the reader decides by the same bins, from the sorted position)
                const bool with_dir = !dir_some || ((p.dir_bins >> (skey & ((1 << BIN_BITS) - 1))) & 1ull);
                const bool coded_n = ntab_some && ((p.ntab_bins >> (skey & ((1 << BIN_BITS) - 1))) & 1ull);
                typedef float quad __attribute__((ext_vector_type(4)));
                const uint32_t gi16 = gi4 << 2;
                quad A, B;
                A.x = rf[0 * TILE + tid]; A.y = rf[1 * TILE + tid]; A.z = rf[2 * TILE + tid]; A.w = rf[R_PIX * TILE + tid];
                B.x = rf[6 * TILE + tid]; B.y = rf[7 * TILE + tid]; B.z = rf[8 * TILE + tid]; B.w = rf[R_MG * TILE + tid];
                st_u(reinterpret_cast<quad *>(stage.quadA()), gi16, A);
                st_u(reinterpret_cast<quad *>(stage.quadB()), gi16, B);
                if (with_dir) {
                    quad D;
                    D.x = rf[3 * TILE + tid]; D.y = rf[4 * TILE + tid]; D.z = rf[5 * TILE + tid]; D.w = p.uses_uv ? rf[13 * TILE + tid] : 0.f;
                    st_u(reinterpret_cast<quad *>(stage.quadD()), gi16, D);
                }
                if (!coded_n) {
                    quad C;
                    C.x = rf[9 * TILE + tid]; C.y = rf[10 * TILE + tid]; C.z = rf[11 * TILE + tid]; C.w = p.uses_uv ? rf[12 * TILE + tid] : 0.f;
                    st_u(reinterpret_cast<quad *>(stage.quadC()), gi16, C);
                }
                st_u(stage.idx(), gi4, skey);
            } else {
                st_u(stage.idx(), gi4, (int32_t)-1);
            }
            if (tid == 0) tile_np[tile] = npend | ((MODE == 1 && pass1_partial) ? *ccnt << 16 : 0);
        }
        __syncthreads();
        STAMP(4);        // sort through LDS + stage write
        if (MODE == 1 && *qcnt > QCAP - TILE) flushQueue(p, seg, qbuf, qcnt, qbase, tid);      // (uniform: read after a barrier)
    }
#ifdef PT_STAMPS
    if (lane == 0 && p.stamps && (blockIdx.x & 15) == 0)           // (a sample of the workgroups: atomics of all of them on 16 words outlast a short kernel)
        for (int k = 0; k < 16; k++) atomicAdd(&p.stamps[(FIRST ? 0 : 16) + k], st_acc[k]);
#endif
#ifdef PT_WGCLOCK
    const unsigned long long wg_t2 = wall_clock64();
#endif
    if (MODE == 1) {                     // counts belong to MODE 2; what is left in the LDS queue goes out now
        if (*qcnt > 0) flushQueue(p, seg, qbuf, qcnt, qbase, tid);
        return;
    }
    // Tail of the sort ("local move").  run_all / run_scat now hold this chunk's survivors / stored paths per bin.  The chunk's
    // stored paths get their place in (bin, tile, rank) order INSIDE the chunk -- cbb[bin] = stored paths of the chunk in earlier
    // bins, + the tile's prefix inside the chunk, + the rank in the tile -- and at that place of the chunk's region of the local
    // index go the path's stage slot and its rank among all survivors of its bin in the chunk.  Nothing of another workgroup is
    // needed, so there is no wait; what IS global (the position of a run in the whole stream) the next launch derives from the
    // run table below.
    int32_t *cbb = toff;                                              // (free after the tile loop)
    if (nb <= 64) {
        if (wave == 0) {
            const int v = lane < nb ? run_scat[lane] : 0;
            const int inc = waveInclusiveScan(v, lane);
            if (lane < nb) cbb[lane] = inc - v;
        }
    } else if (tid == 0) {
        int o = 0;
        for (int b = 0; b < nb; b++) { cbb[b] = o; o += run_scat[b]; }
    }
    __syncthreads();
    for (int b = tid; b < nb; b += TILE) {
        const int ca = run_all[b], cs = run_scat[b];
        const size_t r = (size_t)b * gridDim.x + blockIdx.x;
        chunk_out[r] = ca;
        chunk_out[(size_t)p.chunk_cap + r] = cs;
        chunk_out[2 * (size_t)p.chunk_cap + r] = tile0 * TILE + cbb[b];
        if (ca) { atomicAdd(&super_all[b * p.nsuper + (blockIdx.x >> 6)], ca); atomicAdd(&totals_all[b], ca); }
        if (cs) { atomicAdd(&super_scat[b * p.nsuper + (blockIdx.x >> 6)], cs); atomicAdd(&totals_scat[b], cs); }
    }
    {
        const PathSoA stage = soa_fresh(stage_k);
        const int32_t *keys = stage.idx();
        constexpr int MOVE_U = 4;                                     // tiles per step: their keys are requested before the first is used
        // (only the slots that hold a record are read: the tile's stored count, left by whoever finished the tile -- requested one step
        // ahead, so that the keys stay ONE round trip per step; the ranking pass of a later bounce wrote a key for every slot itself)
        // (the three-barrier ranking pass of a later bounce wrote a key for every slot of every tile: nothing to look up)
        constexpr bool NP_ALL = MODE == 2 && !FIRST && !PT_RANK_ONE_BARRIER;
        int np_next[MOVE_U];                                          // (stored paths at the bottom of the tile | parked rays at its top << 16)
#pragma unroll
        for (int u = 0; u < MOVE_U; u++) np_next[u] = tile0 + u < tile1 ? (NP_ALL ? TILE : tile_np[tile0 + u]) : 0;
        for (int tbase = tile0; tbase < tile1; tbase += MOVE_U) {
            int32_t key[MOVE_U];
            int np_cur[MOVE_U];
#pragma unroll
            for (int u = 0; u < MOVE_U; u++) {
                np_cur[u] = np_next[u];
                np_next[u] = tbase + MOVE_U + u < tile1 ? (NP_ALL ? TILE : tile_np[tbase + MOVE_U + u]) : 0;
            }
#pragma unroll
            for (int u = 0; u < MOVE_U; u++)
                key[u] = (tid < (np_cur[u] & 0xffff) || tid >= TILE - (np_cur[u] >> 16)) ? ld_u(keys, (uint32_t)((tbase + u) * TILE + tid) << 2) : -1;
#pragma unroll
            for (int u = 0; u < MOVE_U; u++) {
                if (key[u] == -1) continue;
                const int tile = tbase + u;
                const int bin = key[u] & ((1 << BIN_BITS) - 1), r_all = (key[u] >> BIN_BITS) & (TILE - 1), r_scat = (int)((uint32_t)key[u] >> (BIN_BITS + RANK_BITS));
                const uint32_t c4 = (uint32_t)(bin * p.maxTiles + tile) << 2;      // (a segment's table is below 4 GiB: ptx_create)
                const int pos = tile0 * TILE + cbb[bin] + ld_u(counts_scat, c4) + r_scat;
                const int slot = tile * TILE + tid, rank = ld_u(counts_all, c4) + r_all;
                if (p.idx16) st_u(stage.lsrc(), (uint32_t)pos << 2, (int32_t)(((uint32_t)(slot - pos) & 0xffffu) | ((uint32_t)rank << 16)));
                else {
                    st_u(stage.lsrc(), (uint32_t)pos << 2, (int32_t)slot);
                    st_u(stage.lidx(), (uint32_t)pos << 2, (int32_t)rank);
                }
            }
        }
    }
#ifdef PT_WGCLOCK
    // (diagnostic build -DPT_WGCLOCK: a slot of its own per (kind, segment of the first 64, workgroup) -- plain adds, no shared address)
    if (tid == 0 && p.stamps && tile1 > tile0 && seg < 64 && blockIdx.x < 4096) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const unsigned long long wg_t3 = wall_clock64();
        unsigned long long *w = p.stamps + 48 + ((size_t)((FIRST ? 0 : 1) * 64 + seg) * 4096 + blockIdx.x) * 5;
        w[0] += wg_t1 - wg_t0; w[1] += wg_t2 - wg_t1; w[2] += wg_t3 - wg_t2; w[3] += 1ull; w[4] += (unsigned long long)(tile1 - tile0);
    }
#endif
}

// Split mesh search, middle part (round 4: refilling waves).  The queue holds one entry per PARKED ray (its stage slot); which
// meshes' boxes the ray reaches travels with it as a bit per geom.  The search of one ray is a walk of unpredictable length -- 9 wide
// nodes and 6 triangles on average on the 20 448-triangle stand-in, the longest of 64 consecutive rays 3-5 times that -- and rounds 2-3
// gave every lane ONE ray: a wave then issues for its slowest lane, 19 % of the lanes active per vector instruction (round 3's
// counters; tools/mesh_walk_sim.cpp reproduces 16 % on the CPU from the walks' step sequences).  Ordering the queue by direction
// octant / entry cell so that neighbouring lanes walk alike buys 5-10 % (same simulator: the spread is in the LENGTHS, not in the
// paths).  So a wave now keeps its lanes busy instead: it draws chunks of the queue (one global atomic per PT_MESHQ_CHUNK entries and
// wave, on a per-segment cursor), every lane holds the walk state of one ray (WideWalk, pt_device.h), and the wave alternates
//   * a NODE round  -- the lanes whose walk holds an inner node do one four-wide node step -- while at least PT_MESH_NMIN lanes do, or
//                      no lane holds a leaf;
//   * a LEAF round  -- the lanes that hold a leaf test ONE of its triangles;
//   * a TURNOVER    -- once PT_MESH_REFILL lanes have nothing to walk (or nothing else is left to do): walks that ended fold their
//                      key into the ray's (minimum of the meshes' keys and the key pass 1 left: cubes and spheres), rays with another
//                      candidate mesh set up its walk, finished rays store their key, free lanes take the next queue entries.
// Same steps on the same data per ray -- wideNodeStep / wideLeafStep are what bvhNearestWide runs -- so the same keys (every mesh test
// green on either schedule); per 64 rays the simulator counts 7.0 k instead of 16.7 k instruction slots (with 220 per turnover and 25
// per round of scheduling), 42 % of the lanes active.  The finishing of the rays (hit decode, terminal cases, record) is k_finish
// again, one dense lane per queue entry: its code and registers do not ride along with the walks.
// Exit: every wave ends when the cursor has passed the queue's end and none of its lanes holds a ray -- each round advances every lane
// it runs, each turnover consumes queue entries or retires rays, so the loop ends for any queue content (bad entries are fenced).
#ifndef PT_MESHQ_CHUNK      // (not PT_MESH_CHUNK: that is pt_device.h's faces-per-lane of the small meshes -- the first build of this kernel
                           // took ITS value, 4, for the chunk: one global atomic per four rays, 8x slower, results right)
#define PT_MESHQ_CHUNK 128    // queue entries a wave reserves per global atomic (64: +1 %, 32: +30 % -- the cursor is one address per segment)
#endif
#ifndef PT_MESH_REFILL
#define PT_MESH_REFILL 16     // lanes without a walk that trigger a turnover
#endif
#ifndef PT_MESH_NMIN
#define PT_MESH_NMIN 32       // lanes holding an inner node that make the next round a node round
#endif
#ifndef PT_MESH_WG_PER_CU
#define PT_MESH_WG_PER_CU 3    // workgroups per CU of k_mesh's grid (see enqueue_batch)
#endif
#ifndef PT_MESH_ONE_TRI
#define PT_MESH_ONE_TRI 0     // 1: a leaf round tests ONE triangle of the leaf in hand; 0: all of them (measured: 0.54 against 0.61 ms per
                              // iteration at 4K -- the kernel waits for memory more than it issues, and a leaf's triangles share cache lines)
#endif
constexpr int MESH_GEOM_WORDS = 20;      // per geom in k_mesh's LDS: inverseTransform rows 0-2 (12), root box lo / hi (6), wide root (-1: not searched here), pad
template <bool FIRST>
__global__ __launch_bounds__(256, PT_MESH_WAVES) void k_mesh(const BounceParams p_in, int bvh_stack) {
    const BounceParams &p = p_in;
    const int seg = blockIdx.y;
    const int n = p.item_count[seg];
    const int lane = threadIdx.x & 63;
    const PathSoA st = soa_offset(p.stage, p.seg_stage * seg);
    const uint32_t *items = p.items + p.seg_items * seg;
    unsigned long long *keys = p.keys + p.seg_keys * seg;
    int32_t *cursor = p.item_cursor + seg;
    const uint32_t slots = p.fence_slots;
    const int ngeoms = p.sc.ngeoms < 32 ? p.sc.ngeoms : 32;
    const uint32_t geom_mask = ngeoms >= 32 ? 0xffffffffu : (1u << ngeoms) - 1u;
    // dynamic LDS: [bvh_stack][256] the walks' stacks, then what a walk's set-up needs per geom -- one LDS read where the geom table, the
    // three per-geom tree tables and the root node were a chain of dependent global loads in front of every walk
    int32_t *stack = pt_lds + threadIdx.x;
    float *gl = reinterpret_cast<float *>(pt_lds + (size_t)bvh_stack * 256);
    if ((int)threadIdx.x < ngeoms) {
        const int gi = threadIdx.x;
        float *o = gl + gi * MESH_GEOM_WORDS;
        const float *G = p.sc.gtab + gi * GTAB_WORDS;
        for (int k = 0; k < 12; k++) o[k] = G[k];
        const int root = p.sc.bvh_root ? p.sc.bvh_root[gi] : -1;
        const bool wideok = root >= 0 && p.sc.bvh_wroot && p.sc.bvh_wroot[gi] >= 0 && p.sc.bvh_wneed[gi] <= bvh_stack;
        BvhQuad A, B;
        A.x = A.y = A.z = B.x = B.y = B.z = 0.f; A.w = B.w = 0;
        if (wideok) { A = p.sc.bvh_nodes[2 * root]; B = p.sc.bvh_nodes[2 * root + 1]; }
        o[12] = A.x; o[13] = A.y; o[14] = A.z; o[15] = B.x; o[16] = B.y; o[17] = B.z;
        o[18] = __int_as_float(wideok ? p.sc.bvh_wroot[gi] : -1); o[19] = 0.f;
    }
    __syncthreads();
    constexpr int32_t IDLE = (int32_t)0x80000001;             // (no leaf reference looks like this either: count 0)
    // per-lane state: the ray in hand (sa: its stage slot; < 0: none), the meshes still to search, the ones left to k_finish, the best
    // key so far, the walk
    int32_t sa = -1, g = 0;
    uint32_t mask = 0, rest = 0;
    unsigned long long key = KEY_NONE;
    WideWalk w;
    w.n = IDLE; w.sp = 0; w.tmin = 0.f; w.face = -1; w.b0 = w.b1 = 0.f;
    w.o = w.d = V3(0.f, 0.f, 0.f); w.ix = w.iy = w.iz = w.enx = w.eny = w.enz = w.efx = w.efy = w.efz = 0.f;
    int cur = 0, end = 0;                                    // (wave-uniform) the chunk of the queue this wave is drawing from
    bool more = n > 0;                                       // (wave-uniform) the queue may still hold entries for this wave
    for (;;) {
        const int n_node = __popcll(__ballot(w.n >= 0)), n_done = __popcll(__ballot(w.n == WIDE_DONE)), n_idle = __popcll(__ballot(w.n == IDLE));
        const int n_leaf = 64 - n_node - n_done - n_idle;
        // lanes a turnover would retire or give a walk: walks that ended, and -- while the queue still has entries -- lanes without a ray
        const int n_wait = more ? n_done + n_idle : n_done;
        if (n_node + n_leaf == 0 || n_wait >= PT_MESH_REFILL) {
            // ---- turnover -----------------------------------------------------------------------------------------------------
            if (w.n == WIDE_DONE) {                          // a walk ended: its key (meshKey's packing: object-space distance, geom, face)
                const float t = w.face >= 0 ? w.tmin : -1.f;
                if (t > 0.0f && t < 3.402823466e+38f) { const unsigned long long km = packKey(t, g, (uint32_t)w.face); key = km < key ? km : key; }
                w.n = IDLE;
            }
            if (more) {                                      // free lanes take the next queue entries
                const unsigned long long m_free = __ballot(w.n == IDLE && !mask);      // (no walk, no mesh left: the ray in hand, if any, retires below)
                if (cur >= end && m_free) {
                    int b = 0;
                    if (lane == 0) b = atomicAdd(cursor, PT_MESHQ_CHUNK);
                    cur = __builtin_amdgcn_readfirstlane(b);
                    end = min(cur + PT_MESHQ_CHUNK, n);
                    if (cur >= n) { more = false; cur = end = 0; }
                }
                const int take = min(__popcll(m_free), end - cur);
                const int mine = wavePrefix(m_free, lane);
                if (w.n == IDLE && !mask) {
                    if (sa >= 0) {                           // every candidate mesh of the ray in hand is searched (or left to k_finish): its key is final here
                        keys[sa] = key;
                        st.nx()[sa] = __int_as_float((int)rest);
                        sa = -1;
                    }
                    if (mine < take) {
                        const uint32_t e = items[cur + mine];
                        if (e < slots) {                     // (fence: a queue entry is a slot of the stage, whatever wrote it)
                            sa = (int32_t)e;
                            key = keys[sa];
                            mask = (uint32_t)__float_as_int(st.nx()[sa]) & geom_mask;
                            rest = 0;
                        } else fence_report(p);
                    }
                }
                cur += take;
            } else if (w.n == IDLE && !mask && sa >= 0) {
                keys[sa] = key;
                st.nx()[sa] = __int_as_float((int)rest);
                sa = -1;
            }
            if (sa >= 0 && w.n == IDLE) {                    // (mask != 0 here) set up the walk of the ray's next candidate mesh
                const vec3 ro = V3(st.px()[sa], st.py()[sa], st.pz()[sa]), rd = V3(st.dx()[sa], st.dy()[sa], st.dz()[sa]);
                while (mask && w.n == IDLE) {
                    g = __ffs((int)mask) - 1;
                    mask &= mask - 1;
                    const float *L = gl + g * MESH_GEOM_WORDS;
                    const int wroot = __float_as_int(L[18]);
                    if (wroot >= 0) {
                        float inv[12];
#pragma unroll
                        for (int k = 0; k < 12; k++) inv[k] = L[k];
                        // (= multiplyMV(geom.inverseTransform, ., .) of meshTestCore: same products, same sums)
                        const vec3 qo = mulRows(inv, ro, 1.0f), qd = normalize(mulRows(inv, rd, 0.0f));
                        BvhQuad A, B;
                        A.x = L[12]; A.y = L[13]; A.z = L[14]; A.w = 0; B.x = L[15]; B.y = L[16]; B.z = L[17]; B.w = 0;
                        wideStart(w, A, B, wroot, qo, qd);
                        if (w.n == WIDE_DONE) w.n = IDLE;    // the root box is missed: no key from this mesh, on to the next
                    } else rest |= 1u << g;                  // a mesh without a four-wide tree (too small for one, or its walk would not fit the
                                                             // stack): k_finish searches it, with the loop or the stackless walk
                }
            }
            if (!more && !__ballot(sa >= 0)) break;
            continue;
        }
        if (n_node >= PT_MESH_NMIN || n_leaf == 0) {
            if (w.n >= 0) wideNodeStep(w, p.sc.bvh_wide, stack, 256);
        } else {
            if (w.n != WIDE_DONE && w.n != IDLE && w.n < 0) wideLeafStep<PT_MESH_ONE_TRI != 0>(w, p.sc.bvh_tris, stack, 256);
        }
    }
}

// Split mesh search, last part: one lane per parked ray finishes it -- nearest hit decoded from the final key, terminal cases, the
// record completed in the slot the ray was parked in if it goes on, and the one word pass 2 needs left in lsrc[owner].  Dense lanes
// that all do the same thing: the dependent face / texel loads of a textured mesh hit hide behind the other waves.
template <bool FIRST>
__global__ __launch_bounds__(256) void k_finish(const BounceParams p_in) {
    BounceParams p = p_in;
    p.sc.tri_lds = 0; p.sc.ntri_lds = 0;
    const int seg = blockIdx.y;
    const int iter = p.iter + seg * p.iter_stride;
    const int n = p.item_count[seg];
    const PathSoA st = soa_offset(p.stage, p.seg_stage * seg);
    const uint32_t *items = p.items + p.seg_items * seg;
    const unsigned long long *keys = p.keys + p.seg_keys * seg;
    float *part = p.part ? p.part + p.seg_part * seg : nullptr;
    const bool batched = part != nullptr;
    const uint32_t slots = p.fence_slots;
    const uint32_t geom_mask = p.sc.ngeoms >= 32 ? 0xffffffffu : (1u << p.sc.ngeoms) - 1u;
    for (int k = blockIdx.x * blockDim.x + threadIdx.x; k < n; k += gridDim.x * blockDim.x) {
        const int sa = (int)items[k];
        if ((uint32_t)sa >= slots) { fence_report(p); continue; }      // (fence: a queue entry is a slot of the stage, whatever wrote it)
        Ray ray;
        ray.o = V3(st.px()[sa], st.py()[sa], st.pz()[sa]);
        ray.d = V3(st.dx()[sa], st.dy()[sa], st.dz()[sa]);
        unsigned long long key = keys[sa];
        // the candidate meshes k_mesh did not search (no four-wide tree: small meshes of a split scene, trees too deep for the walks' stack):
        // the plain loop or the stackless walk, same keys, same minimum
        for (uint32_t m = (uint32_t)__float_as_int(st.nx()[sa]) & geom_mask; m; m &= m - 1) {
            const unsigned long long km = meshKey(p.sc, p.sc.gtab, __ffs((int)m) - 1, ray);
            key = km < key ? km : key;
        }
        const int owner = st.mg()[sa], pix = st.pix()[sa];
        if ((uint32_t)owner >= slots || (uint32_t)pix >= (uint32_t)p.tm.owned) { fence_report(p); continue; }
        const vec3 color = V3(st.cr()[sa], st.cg()[sa], st.cb()[sa]);
        Hit hit;
        decodeKey(p.sc, p.sc.gtab, key, ray, p.uses_uv != 0, hit);
        int bin = 0;
        bool pending = false;
        classifyPath<FIRST>(p, iter, part, batched, hit, color, pix, bin, pending);
        if (pending) {
            const vec3 sp = add(ray.o, scale(ray.d, hit.t));          // the point shadeFakeMaterial will shade (:392)
            st.px()[sa] = sp.x; st.py()[sa] = sp.y; st.pz()[sa] = sp.z;
            // (direction, colour and pixel are in place; a record of a cubes-only material carries its normal as a code in the pixel word)
            if (p.nbins <= 64 && ((p.ntab_bins >> bin) & 1ull)) st.pix()[sa] = pix | (hit.ncode << 28);
            else { st.nx()[sa] = hit.n.x; st.ny()[sa] = hit.n.y; st.nz()[sa] = hit.n.z; }
            if (p.uses_uv) { st.u()[sa] = hit.u; st.v()[sa] = hit.v; }
            st.mg()[sa] = hit.mat | (hit.geom << 16);
        }
        else if (PT_RANK_ONE_BARRIER) st.idx()[sa] = -1;           // (that form's tail reads the keys of every parked ray's slot: none here)
        st.lsrc()[owner] = K1_ALIVE | (pending ? K1_PEND : 0) | bin | ((sa & (TILE - 1)) << 16);
    }
}

// debug capture: the sorted stream materialised -- what the next k_bounce would read, resolved the slow, obvious way from the same
// tables (run prefixes by one thread, a binary search per position), so that the parity tests see the order the kernels define
// without going through k_bounce's own window search.
__global__ void k_capture_prefix(const int32_t *chunk, int chunk_cap, int nruns, int32_t *gs, int32_t *ga) {
    if (blockIdx.x != 0 || threadIdx.x != 0) return;
    int s = 0, a = 0;
    for (int r = 0; r < nruns; r++) { gs[r] = s; ga[r] = a; a += chunk[r]; s += chunk[chunk_cap + r]; }
    gs[nruns] = s; ga[nruns] = a;
}
__global__ void k_capture(PathSoA stage, const int32_t *chunk, int chunk_cap, int nruns, const int32_t *gs, const int32_t *ga, int cap,
                          int32_t *out_i, float *out_f, int idx16) {
    const int n = min(gs[nruns], cap);
    for (int k = blockIdx.x * blockDim.x + threadIdx.x; k < n; k += gridDim.x * blockDim.x) {
        int lo = 0, hi = nruns - 1;                              // last run that starts at or before k
        while (lo < hi) {
            const int mid = (lo + hi + 1) >> 1;
            if (gs[mid] <= k) lo = mid; else hi = mid - 1;
        }
        const int li = chunk[2 * chunk_cap + lo] + (k - gs[lo]);
        const uint32_t w = (uint32_t)stage.lsrc()[li];
        int j = idx16 ? li + (int)(int16_t)(w & 0xffffu) : (int)w;
        const int rank = idx16 ? (int)(w >> 16) : stage.lidx()[li];
        j = j < 0 ? 0 : (j >= cap ? cap - 1 : j);
        out_i[k] = stage.pix()[j]; out_i[(size_t)cap + k] = ga[lo] + rank; out_i[2 * (size_t)cap + k] = stage.mg()[j];
        for (int f = 0; f < SOA_LOGICAL_FLOATS; f++) out_f[(size_t)f * cap + k] = stage.fieldAt(f, (size_t)j);
    }
}

// replay of the cached bounce-0 light hits (first-bounce cache, iterations > 1)
// add != 0: image[pix] += rgb (one iteration at a time); add == 0: store into the per-iteration radiance buffer of each of
// the nseg segments (batched mode; the buffers were cleared, so bounce-0 misses read as 0)
__global__ void k_replay_emission(TileMap tm, const int32_t *count, const int32_t *pix, const float *rgb, float *dst, size_t seg_stride,
                                  int nseg, int add, uint32_t cap) {
    int n = *count;
    for (int k = blockIdx.x * blockDim.x + threadIdx.x; k < n; k += gridDim.x * blockDim.x) {
        const float r = rgb[k * 3 + 0], g = rgb[k * 3 + 1], b = rgb[k * 3 + 2];
        if (add) {
            float *px = dst + (size_t)slot_to_pixel(tm, pix[k]) * 3;      // dst = the image (pixels), pix[] holds slots
            px[0] += r; px[1] += g; px[2] += b;
        } else {
            for (int sg = 0; sg < nseg; sg++) {
                float *px = dst + seg_stride * sg + (size_t)pix[k] * 3;
                px[0] = r; px[1] = g; px[2] = b;
                lit_flags(dst + seg_stride * sg, cap)[pix[k]] = 1;
            }
        }
    }
}

// the cached bounce-0 totals into the totals block of every segment of a batch
__global__ void k_seed_totals(int32_t *dst, size_t seg_totals, int nseg, const int32_t *src, int n) {
    for (int k = threadIdx.x; k < n * nseg; k += blockDim.x) dst[seg_totals * (k / n) + (k % n)] = src[k % n];
}

// batched mode: image[pix] += part[0][pix]; += part[1][pix]; ... in iteration order, over the pixels this device owns
// (only the slots whose "lit" flag of that iteration is set hold anything: see deposit)
__global__ void k_gather(TileMap tm, int resx, int nseg, size_t seg_part, const float *part, float *image, uint32_t cap) {
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < tm.owned; i += gridDim.x * blockDim.x) {
        int x, y;
        owned_pixel(tm, i, x, y);
        const size_t o = ((size_t)x + (size_t)y * resx) * 3;
        unsigned long long lit = 0;                      // bit s: iteration s of the batch ended this pixel's path on a light (nseg <= 64)
        for (int s = 0; s < nseg; s++) lit |= (unsigned long long)(reinterpret_cast<const uint8_t *>(part + seg_part * s + 3 * (size_t)cap)[i] != 0) << s;
        if (!lit) continue;                              // (a pixel no iteration of the batch lit: its sum does not move)
        float r = image[o], g = image[o + 1], b = image[o + 2];
        for (; lit; lit &= lit - 1) {                    // in iteration order: the same fp32 sums as one iteration at a time
            const float *ps = part + seg_part * (size_t)(__ffsll((long long)lit) - 1) + (size_t)i * 3;
            r += ps[0]; g += ps[1]; b += ps[2];
        }
        image[o] = r; image[o + 1] = g; image[o + 2] = b;
    }
}

// per-iteration statistics: rays entering the intersect stage of each bounce = sum of totals_all[bounce]
// (bounce 0 is not counted on iterations that took it from the first-bounce cache: nothing was traced -- so the cached bounce-0 records
// that every such iteration RE-READS are not part of stored_* either: the mix describes what was written, once.)
// dir_bins / ntab_bins = the masks the batch's launches wrote with (enqueue_batch: batch_dir_bins), not the tracer's.
__global__ void k_stats(const int32_t *totals, int nbins, int nbounces, int stride, int skip_first, int nseg, size_t seg_totals,
                        int64_t *last, int64_t *total, unsigned long long dir_bins, unsigned long long ntab_bins, int64_t *kinds) {
    // one wave: lane j takes the (segment, bounce) pairs j, j + 64, ...
    if (blockIdx.x != 0 || threadIdx.x >= 64) return;
    long long sum = 0, st = 0, sd = 0, sn = 0;      // rays; stored paths: all, with a direction, with a normal code (what the records weigh)
    for (int k = threadIdx.x; k < nseg * nbounces; k += 64) {
        const int sg = k / nbounces, b = k - sg * nbounces;
        long long s = 0;
        if (!(b == 0 && skip_first))
            for (int q = 0; q < nbins; q++) {
                s += totals[seg_totals * sg + (size_t)b * stride + q];
                const long long c = totals[seg_totals * sg + (size_t)b * stride + nbins + q];
                st += c;
                if (nbins > 64 || ((dir_bins >> q) & 1ull)) sd += c;
                if (nbins <= 64 && ((ntab_bins >> q) & 1ull)) sn += c;
            }
        if (sg == nseg - 1 && b < 64) last[b] = s;      // per-bounce counts of the last iteration of the batch
        sum += s;
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) { sum += __shfl_xor(sum, off); st += __shfl_xor(st, off); sd += __shfl_xor(sd, off); sn += __shfl_xor(sn, off); }
    if (threadIdx.x == 0) {
        atomicAdd(reinterpret_cast<unsigned long long *>(total), (unsigned long long)sum);
        atomicAdd(reinterpret_cast<unsigned long long *>(kinds), (unsigned long long)st);
        atomicAdd(reinterpret_cast<unsigned long long *>(kinds + 1), (unsigned long long)sd);
        atomicAdd(reinterpret_cast<unsigned long long *>(kinds + 2), (unsigned long long)sn);
    }
}

// sendImageToPBO, src/pathtrace.cu:69-89
__global__ void k_pbo(uchar4 *pbo, int n, int iter, const float *image) {
    int index = blockIdx.x * blockDim.x + threadIdx.x;
    if (index < n) {
        const float *pix = image + (size_t)index * 3;
        int c[3];
        for (int k = 0; k < 3; k++) {
            int v = (int)(pix[k] / (float)iter * 255.0);
            c[k] = v < 0 ? 0 : (v > 255 ? 255 : v);
        }
        uchar4 o; o.w = 0; o.x = (unsigned char)c[0]; o.y = (unsigned char)c[1]; o.z = (unsigned char)c[2];
        pbo[index] = o;
    }
}

// ---- per-stage kernels for the parity tests (AoS records of the reference in, same out) ----------------------
struct HostPath { float o[3], d[3], c[3]; int32_t pixelIndex, remainingBounces; };       // 44 B
struct HostIsect { float t, n[3]; int32_t materialId; float uv[2]; int32_t geomId; };    // 32 B

__global__ void k_kat_geom(DScene sc, int gi, int n, const float *rays, float *out) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const DGeom &g = sc.geoms[gi];
    Ray r; r.o = ld3(rays + i * 6); r.d = ld3(rays + i * 6 + 3);
    vec3 p = V3(0, 0, 0), nrm = V3(0, 0, 0);
    float u = 0.f, v = 0.f;
    bool outside = true;
    float t = -1.f;
    if (g.type == G_CUBE) t = boxIntersectionTest(g, r, p, nrm, outside);
    else if (g.type == G_SPHERE) t = sphereIntersectionTest(g, r, p, nrm, outside);
    else if (g.type == G_OBJ) t = meshIntersectionTest(sc, g, r, p, nrm, u, v, outside, sc.bvh_root ? sc.bvh_root[gi] : -1);
    float *o = out + i * 10;
    o[0] = t; o[1] = p.x; o[2] = p.y; o[3] = p.z; o[4] = nrm.x; o[5] = nrm.y; o[6] = nrm.z; o[7] = u; o[8] = v;
    o[9] = outside ? 1.f : 0.f;
}

// the reference's dead objTriIntersectionTest (src/intersections.h:284-315) on an OBJ geom: out per ray = t, point, normal, outside
__global__ void k_kat_obj_tri(DScene sc, int gi, int n, const float *rays, float *out) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const DGeom &g = sc.geoms[gi];
    Ray r; r.o = ld3(rays + i * 6); r.d = ld3(rays + i * 6 + 3);
    vec3 p = V3(0, 0, 0), nrm = V3(0, 0, 0);
    bool outside = true;
    const float t = g.type == G_OBJ ? objTriTest(sc, g, r, p, nrm, outside) : -1.f;
    float *o = out + i * 8;
    o[0] = t; o[1] = p.x; o[2] = p.y; o[3] = p.z; o[4] = nrm.x; o[5] = nrm.y; o[6] = nrm.z; o[7] = outside ? 1.f : 0.f;
}

// the reference's dead calculateJitteredDirectionHemisphere (src/interactions.h:46-85): normal + (iter, index, depth) -> direction
__global__ void k_kat_jittered(int n, const float *normals, const int32_t *seeds, int max_iter, float *out) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    Rng rng; rng.seed(seeds[i * 3], seeds[i * 3 + 1], seeds[i * 3 + 2]);
    const vec3 d = jitteredDirectionInHemisphere(ld3(normals + i * 3), rng, seeds[i * 3], max_iter);
    out[i * 3] = d.x; out[i * 3 + 1] = d.y; out[i * 3 + 2] = d.z;
}

__global__ void k_kat_intersect(DScene sc, int n, const HostPath *paths, HostIsect *out) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    Ray r; r.o = ld3(paths[i].o); r.d = ld3(paths[i].d);
    Hit h;
    intersectScene(sc, r, h);
    HostIsect o;
    memset(&o, 0, sizeof o);
    if (h.t > 0.f) {
        o.t = h.t; o.n[0] = h.n.x; o.n[1] = h.n.y; o.n[2] = h.n.z; o.materialId = h.mat; o.uv[0] = h.u; o.uv[1] = h.v;
        o.geomId = h.geom;
    } else {
        o.t = -1.f;
    }
    out[i] = o;
}

// computeIntersections as PRODUCTION runs it, on arbitrary rays: candidate masks from the world boxes (cullMask), the tile's (ray, geom)
// pairs pooled in LDS and tested by primKey / meshKey, 64-bit LDS minimum, winner decoded by decodeKey -- tileIntersect itself, with
// the scene tables staged as k_bounce stages them.  SPLIT: the three pieces of the split mesh search instead -- tileIntersect<DEFER>
// (pass 1), meshKey with the stack traversal for every mesh of the ray's candidate mask, folded in by minimum, and decodeKey (k_mesh).
// A named test for the functions that the frame-level parity tests only reach through whole bounces.
template <bool SPLIT>
__global__ __launch_bounds__(TILE) void k_kat_tile(DScene sc, DScene scg, int n, const HostPath *paths, HostIsect *out, int uses_uv) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int32_t *lds = pt_lds + sceneLdsWords(sc);
    int32_t *tcnt = lds;                                             // [8] (the rest of the head is unused here)
    int32_t *rec = lds + ldsHeadWords(1);
    int32_t *stack = rec + REC_WORDS;                                // SPLIT: [bvh_stack][TILE]
    stageSceneToLds(sc, tid, TILE);
    if (tid < 8) tcnt[tid] = 0;
    __syncthreads();
    int tq = 0;
    const float *gtab_lds = reinterpret_cast<const float *>(pt_lds) + sc.ntri_lds * 24 + sc.nmats * 11;
    for (int base = blockIdx.x * TILE; base < n; base += gridDim.x * TILE) {
        const int i = base + tid;
        const bool alive = i < n;
        Ray ray; ray.o = ray.d = V3(0.f, 0.f, 0.f);
        if (alive) { ray.o = ld3(paths[i].o); ray.d = ld3(paths[i].d); }
        Hit h;
        h.t = -1.f; h.n = V3(0.f, 0.f, 0.f); h.u = h.v = 0.f; h.geom = 0; h.mat = 0;
        unsigned long long key = KEY_NONE;
        uint32_t mesh_cand = 0;
#ifdef PT_STAMPS
        unsigned long long st_acc[16] = {0}, st_t0 = 0;        // (the phase-timing build: this kernel's stamps go nowhere)
#endif
        if (!SPLIT) {
            tileIntersect<false>(sc, alive, ray, uses_uv != 0, h, rec, tcnt, tq, tid, lane, wave, key, mesh_cand TI_PASS);
        } else {
            tileIntersect<true>(sc, alive, ray, uses_uv != 0, h, rec, tcnt, tq, tid, lane, wave, key, mesh_cand TI_PASS);
            for (uint32_t m = mesh_cand; m; m &= m - 1) {
                const unsigned long long k = meshKey(scg, scg.gtab, __ffs((int)m) - 1, ray, -1, stack + tid, TILE);
                key = k < key ? k : key;
            }
            if (alive) decodeKey(sc, gtab_lds, key, ray, uses_uv != 0, h);
        }
        if (alive) {
            HostIsect o;
            memset(&o, 0, sizeof o);
            if (h.t > 0.f) { o.t = h.t; o.n[0] = h.n.x; o.n[1] = h.n.y; o.n[2] = h.n.z; o.materialId = h.mat; o.uv[0] = h.u; o.uv[1] = h.v; o.geomId = h.geom; }
            else o.t = -1.f;
            out[i] = o;
        }
        __syncthreads();                                             // (tileIntersect's scratch is reused by the next tile)
        __syncthreads();
    }
}

// shadeFakeMaterial in full (src/pathtrace.cu:365-403), one path per thread, idx[] = RNG stream indices
__global__ void k_kat_shade(DScene sc, int iter, int n, const int32_t *idx, const HostIsect *isects, HostPath *paths) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    HostIsect is = isects[i];
    HostPath seg = paths[i];
    if (is.t > 0.0f) {
        const DMaterial &m = sc.mats[is.materialId];
        if (m.emittance > 0.0f) {
            vec3 c = mul(ld3(seg.c), scale(V3(m.color[0], m.color[1], m.color[2]), m.emittance));
            seg.c[0] = c.x; seg.c[1] = c.y; seg.c[2] = c.z;
            seg.remainingBounces = 0;
        } else if (seg.remainingBounces == 1) {
            seg.c[0] = seg.c[1] = seg.c[2] = 0.f;
            seg.remainingBounces = 0;
        } else {
            PathState ps; ps.o = ld3(seg.o); ps.d = ld3(seg.d); ps.color = ld3(seg.c);
            Hit h; h.t = is.t; h.n = ld3(is.n); h.u = is.uv[0]; h.v = is.uv[1]; h.mat = is.materialId; h.geom = is.geomId;
            Rng rng; rng.seed(iter, idx[i], 0);
            vec3 intersect = add(ps.o, scale(ps.d, h.t));
            bool ended = scatterRay(sc, ps, intersect, h, m, rng);
            seg.o[0] = ps.o.x; seg.o[1] = ps.o.y; seg.o[2] = ps.o.z;
            seg.d[0] = ps.d.x; seg.d[1] = ps.d.y; seg.d[2] = ps.d.z;
            seg.c[0] = ps.color.x; seg.c[1] = ps.color.y; seg.c[2] = ps.color.z;
            if (ended) seg.remainingBounces = 1;
            seg.remainingBounces -= 1;
        }
    } else {
        seg.c[0] = seg.c[1] = seg.c[2] = 0.f;
        seg.remainingBounces = 0;
    }
    paths[i] = seg;
}

__global__ void k_kat_generate(DCamera cam, int iter, int traceDepth, int aa, int dof, HostPath *paths) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    int n = cam.resx * cam.resy;
    if (i >= n) return;
    int y = i / cam.resx, x = i - y * cam.resx;
    PathState ps;
    generateRay(cam, iter, traceDepth, aa != 0, dof != 0, x, y, ps);
    HostPath seg;
    seg.o[0] = ps.o.x; seg.o[1] = ps.o.y; seg.o[2] = ps.o.z;
    seg.d[0] = ps.d.x; seg.d[1] = ps.d.y; seg.d[2] = ps.d.z;
    seg.c[0] = ps.color.x; seg.c[1] = ps.color.y; seg.c[2] = ps.color.z;
    seg.pixelIndex = i; seg.remainingBounces = traceDepth;
    paths[i] = seg;
}

__global__ void k_kat_libm(int n, const float *x, float *s, float *c, const double *pw, double *p5,
                           const float *pxy, float *pout) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float sn, cs;
    sincos_pt(x[i], &sn, &cs);          // (the routine the samplers call: sincos_own at the exact level)
    s[i] = sn; c[i] = cs;
    p5[i] = pow5_own(pw[i]);
    pout[i] = powf_own(pxy[2 * i], pxy[2 * i + 1]);
}

// every float bit pattern through the guarded core routines and through the compiler's own expansions
__global__ void k_kat_fast_exact(unsigned long long *mism) {
    unsigned long long m0 = 0, m1 = 0, m2 = 0;
    const unsigned long long total = 1ull << 32, step = (unsigned long long)gridDim.x * blockDim.x;
    for (unsigned long long k = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; k < total; k += step) {
        float x = __uint_as_float((uint32_t)k);
        asm volatile("" : "+v"(x));                          // (the two sides must not be merged into one computation)
        float y = x;
        asm volatile("" : "+v"(y));
        const float a0 = pt_sqrt(x), b0 = __builtin_sqrtf(y);
        const float a1 = pt_rsqrt_glm(x), b1 = 1.0f / __builtin_sqrtf(y);
        // pt_rcp_pos is only ever called with a >= FLT_EPSILON (or NaN): the patterns below that are the caller's business
        const bool dom = !(x < 1.1920928955078125e-07f);
        const float a2 = dom ? pt_rcp_pos(x) : 0.f, b2 = dom ? 1.0f / y : 0.f;
        auto same = [](float p, float q) { return __float_as_uint(p) == __float_as_uint(q) || (p != p && q != q); };
        m0 += same(a0, b0) ? 0 : 1; m1 += same(a1, b1) ? 0 : 1; m2 += same(a2, b2) ? 0 : 1;
    }
    if (m0) atomicAdd(&mism[0], m0);
    if (m1) atomicAdd(&mism[1], m1);
    if (m2) atomicAdd(&mism[2], m2);
}

// ---- the arithmetic-bearing kernels of THIS translation unit as a table of launchers ---------------------------------------------
// The file is compiled once per arithmetic level (pt_device.h: PT_ARITH; levels 1 and 2 through csrc/pt_arith.hip, each a code object
// of its own).  Only the level-0 translation unit holds the host side; it launches through one of these tables, its own or one it
// gets from ptx_arith_kernels_<level>().  Plain types in the signatures (the parameter blocks travel as const void *): the structs are
// the same source in every translation unit but, formally, types of different anonymous namespaces.
struct KernelSet {
    int arith;
    void (*bounce)(int first, int mode, int fast, dim3 grid, size_t lds, hipStream_t st, const void *bounce_params);
    void (*mesh)(int first, dim3 grid, size_t lds, hipStream_t st, const void *bounce_params, int bvh_stack);
    void (*finish)(int first, dim3 grid, hipStream_t st, const void *bounce_params);
    void (*kat_geom)(dim3 grid, hipStream_t st, const void *scene, int gi, int n, const float *rays, float *out);
    void (*kat_intersect)(dim3 grid, hipStream_t st, const void *scene, int n, const void *paths, void *out);
    void (*kat_obj_tri)(dim3 grid, hipStream_t st, const void *scene, int gi, int n, const float *rays, float *out);
    void (*kat_jittered)(dim3 grid, hipStream_t st, int n, const float *normals, const int32_t *seeds, int max_iter, float *out);
    void (*kat_tile)(int split, dim3 grid, size_t lds, hipStream_t st, const void *sc, const void *scg, int n, const void *paths, void *out, int uses_uv);
    void (*kat_shade)(dim3 grid, hipStream_t st, const void *scene, int iter, int n, const int32_t *idx, const void *isects, void *paths);
    void (*kat_generate)(dim3 grid, hipStream_t st, const void *cam, int iter, int traceDepth, int aa, int dof, void *paths);
    void (*kat_libm)(dim3 grid, hipStream_t st, int n, const float *x, float *s, float *c, const double *pw, double *p5, const float *pxy, float *pout);
};

template <bool FIRST, int MODE>
void launch_bounce_variant(bool fast, dim3 grid, size_t lds, hipStream_t stream, const BounceParams &bp) {
    if (fast) hipLaunchKernelGGL((k_bounce<FIRST, MODE, true>), grid, dim3(TILE), lds - (MODE == 0 ? sizeof(int32_t) * (17 - REC_ROWS_FAST0) * TILE : 0), stream, bp);
    else hipLaunchKernelGGL((k_bounce<FIRST, MODE, false>), grid, dim3(TILE), lds, stream, bp);
}
void ks_bounce(int first, int mode, int fast, dim3 grid, size_t lds, hipStream_t stream, const void *params) {
    const BounceParams &bp = *static_cast<const BounceParams *>(params);
    if (first) {
        if (mode == 0) launch_bounce_variant<true, 0>(fast != 0, grid, lds, stream, bp);
        else if (mode == 1) launch_bounce_variant<true, 1>(fast != 0, grid, lds, stream, bp);
        else launch_bounce_variant<true, 2>(fast != 0, grid, lds, stream, bp);
    } else {
        if (mode == 0) launch_bounce_variant<false, 0>(fast != 0, grid, lds, stream, bp);
        else if (mode == 1) launch_bounce_variant<false, 1>(fast != 0, grid, lds, stream, bp);
        else launch_bounce_variant<false, 2>(fast != 0, grid, lds, stream, bp);
    }
}
void ks_mesh(int first, dim3 grid, size_t lds, hipStream_t stream, const void *params, int bvh_stack) {
    const BounceParams &bp = *static_cast<const BounceParams *>(params);
    if (first) hipLaunchKernelGGL(k_mesh<true>, grid, dim3(256), lds, stream, bp, bvh_stack);
    else hipLaunchKernelGGL(k_mesh<false>, grid, dim3(256), lds, stream, bp, bvh_stack);
}
void ks_finish(int first, dim3 grid, hipStream_t stream, const void *params) {
    const BounceParams &bp = *static_cast<const BounceParams *>(params);
    if (first) hipLaunchKernelGGL(k_finish<true>, grid, dim3(256), 0, stream, bp);
    else hipLaunchKernelGGL(k_finish<false>, grid, dim3(256), 0, stream, bp);
}
void ks_kat_geom(dim3 grid, hipStream_t st, const void *scene, int gi, int n, const float *rays, float *out) {
    hipLaunchKernelGGL(k_kat_geom, grid, dim3(256), 0, st, *static_cast<const DScene *>(scene), gi, n, rays, out);
}
void ks_kat_obj_tri(dim3 grid, hipStream_t st, const void *scene, int gi, int n, const float *rays, float *out) {
    hipLaunchKernelGGL(k_kat_obj_tri, grid, dim3(256), 0, st, *static_cast<const DScene *>(scene), gi, n, rays, out);
}
void ks_kat_jittered(dim3 grid, hipStream_t st, int n, const float *normals, const int32_t *seeds, int max_iter, float *out) {
    hipLaunchKernelGGL(k_kat_jittered, grid, dim3(256), 0, st, n, normals, seeds, max_iter, out);
}
void ks_kat_intersect(dim3 grid, hipStream_t st, const void *scene, int n, const void *paths, void *out) {
    hipLaunchKernelGGL(k_kat_intersect, grid, dim3(256), 0, st, *static_cast<const DScene *>(scene), n, static_cast<const HostPath *>(paths), static_cast<HostIsect *>(out));
}
void ks_kat_tile(int split, dim3 grid, size_t lds, hipStream_t st, const void *sc, const void *scg, int n, const void *paths, void *out, int uses_uv) {
    const DScene &a = *static_cast<const DScene *>(sc), &b = *static_cast<const DScene *>(scg);
    if (split) hipLaunchKernelGGL(k_kat_tile<true>, grid, dim3(TILE), lds, st, a, b, n, static_cast<const HostPath *>(paths), static_cast<HostIsect *>(out), uses_uv);
    else hipLaunchKernelGGL(k_kat_tile<false>, grid, dim3(TILE), lds, st, a, b, n, static_cast<const HostPath *>(paths), static_cast<HostIsect *>(out), uses_uv);
}
void ks_kat_shade(dim3 grid, hipStream_t st, const void *scene, int iter, int n, const int32_t *idx, const void *isects, void *paths) {
    hipLaunchKernelGGL(k_kat_shade, grid, dim3(256), 0, st, *static_cast<const DScene *>(scene), iter, n, idx, static_cast<const HostIsect *>(isects), static_cast<HostPath *>(paths));
}
void ks_kat_generate(dim3 grid, hipStream_t st, const void *cam, int iter, int traceDepth, int aa, int dof, void *paths) {
    hipLaunchKernelGGL(k_kat_generate, grid, dim3(256), 0, st, *static_cast<const DCamera *>(cam), iter, traceDepth, aa, dof, static_cast<HostPath *>(paths));
}
void ks_kat_libm(dim3 grid, hipStream_t st, int n, const float *x, float *s, float *c, const double *pw, double *p5, const float *pxy, float *pout) {
    hipLaunchKernelGGL(k_kat_libm, grid, dim3(256), 0, st, n, x, s, c, pw, p5, pxy, pout);
}
const KernelSet g_kernels_here = {PT_ARITH, ks_bounce, ks_mesh, ks_finish, ks_kat_geom, ks_kat_intersect, ks_kat_obj_tri, ks_kat_jittered, ks_kat_tile, ks_kat_shade, ks_kat_generate, ks_kat_libm};

}  // namespace

#if PT_ARITH != 0
// this translation unit is one of the extra code objects: all it exports is its table
#define PT_ARITH_EXPORT_(n) ptx_arith_kernels_##n
#define PT_ARITH_EXPORT(n) PT_ARITH_EXPORT_(n)
extern "C" const void *PT_ARITH_EXPORT(PT_ARITH)(void) { return &g_kernels_here; }
#else
// the other arithmetic levels' tables (weak: a library linked without csrc/pt_arith.hip's objects still loads, and refuses arith != 0)
extern "C" const void *ptx_arith_kernels_1(void) __attribute__((weak));
extern "C" const void *ptx_arith_kernels_2(void) __attribute__((weak));

// ---------------------------------------------------------------------------------------------------------------
struct ptx_tracer {
    int device = 0;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    hipEvent_t ev_start = nullptr, ev_stop = nullptr;
    bool timing_valid = false;
    ptx_options opt{};
    DCamera cam{};
    int traceDepth = 0;
    TileMap tm{};
    int nbins = 1, nmats = 0, ngeoms = 0, maxTiles = 0, grid = 0, grid_seg = 0, cap = 0, cus = 0;
    bool grid_forced = false;                  // PTX_DEBUG_WG_PER_CU given: the grid is what it says for every kernel
    int dbg_mesh_wg_per_cu = 0;                 // PTX_DEBUG_MESH_WG_PER_CU: workgroups per CU of k_mesh's grid (tuning experiments)
    int dbg_extra_lds = 0;
    int dbg_total_wg_per_cu = 0, dbg_nsets = 0; // PTX_DEBUG_TOTAL_WG_PER_CU / PTX_DEBUG_NSETS: tuning experiments (grid of a whole launch; sets of a short run)
    // device memory
    DGeom *d_geoms = nullptr; DMaterial *d_mats = nullptr; float *d_faces = nullptr; uint8_t *d_texels = nullptr;
    float *d_image = nullptr; bool own_image = false;
    float *d_fbuf[3] = {nullptr, nullptr, nullptr};      // stream, stage, cache
    int32_t *d_ibuf[3] = {nullptr, nullptr, nullptr};
    PathSoA soa[3];                                      // 0, 1 = the two stages (bounce b writes soa[1 - (b & 1)], b + 1 reads it), 2 = first-bounce cache
    int32_t *d_counts = nullptr;                         // per segment: [2][nbins][maxTiles] prefix tables + [maxTiles] stored paths per tile
    int32_t *d_chunk = nullptr;                          // [segments][2 (bounce parity)][3][nbins x grid_seg]: the run tables (BounceParams::chunk)
    int32_t *d_cache_chunk = nullptr;                    // [3][nbins x grid_seg]: the cached bounce 0's
    int32_t *d_cache_super = nullptr;                    // [2][nbins][nsuper]: the cached bounce 0's
    int cache_gx = 0;                                    // workgroups per segment of the launch that filled the cache (its run tables' width)
    int32_t *d_totals = nullptr;                         // [maxBounces][2][nbins] then [maxBounces][2][nbins][nsuper]
    int32_t *d_super = nullptr;                          // (points into d_totals' allocation)
    float *d_tri9 = nullptr, *d_gtab = nullptr, *d_aabb = nullptr;
    std::vector<float> h_aabb;                           // host copy of the world boxes (update_tile_geoms)
    uint32_t *d_tile_geoms = nullptr; bool tile_geoms_valid = false;      // BounceParams::tile_geoms of the current camera
    uint32_t cube_bits = 0, sphere_bits = 0, mesh_bits = 0;   // geoms 0..31 by kind, for the candidate masks
    BvhQuad *d_bvh_nodes = nullptr; float *d_bvh_tris = nullptr; int32_t *d_bvh_root = nullptr, *d_bvh_depth = nullptr;   // pt_bvh.h (NULL: no mesh has one)
    BvhWide4 *d_bvh_wide = nullptr; int32_t *d_bvh_wroot = nullptr, *d_bvh_wneed = nullptr;                                  // four-wide nodes of the same trees (k_mesh)
    int ntri_lds = 0, bvh_nodes = 0, bvh_meshes = 0, bvh_stack = BVH_STACK;
    int mesh_chunks = 1;                                 // see DScene::mesh_chunks
    float *d_fnorm = nullptr, *d_cnorm = nullptr;        // precomputed normals (DScene::fnorm / cnorm)
    float *d_ldsblob = nullptr;                          // DScene::ldsblob for the ntri_lds k_bounce is launched with
    uint32_t bump_bits = 0;
    bool split_mesh = false;                             // k_bounce as MODE 1 + k_mesh + k_finish + MODE 2 (scenes with BVH meshes)
    bool no_fast = false;                                // PTX_DEBUG_NO_FAST: always the general k_bounce (A/B timing, tests of both variants)
    bool force_fast = false;                             // PTX_DEBUG_FORCE_FAST: ask for the specialised variant at every launch (refused
                                                         // with PTX_ERR_INVALID where its preconditions do not hold; tests only)
    unsigned long long *d_keys = nullptr; uint32_t *d_items = nullptr; int32_t *d_item_count = nullptr;
    int32_t *d_tile_done = nullptr;                      // split first bounce: [segments][maxTiles], see BounceParams::tile_done
    size_t seg_items = 0;
    int cull = 0;
    int nsuper = 1, ntri = 0, tri_lds = 0;
    size_t totals_bytes = 0, seg_totals = 0, field_stride = 0, seg_part = 0;
    int kmax = 1;                                        // iterations per launch set (segments)
    long long split_min_paths = 1LL << 20;               // ptx_render_strided: smallest launch set a short run is cut into
    int lanes = 1;                                       // launch sets in flight at once, each on a stream of its own with its own
                                                         // kmax segments of every per-iteration buffer (lane 0 = `stream`)
    hipStream_t lane_stream[MAX_LANES] = {};      // [0] = `stream`, the others are the tracer's own
    hipEvent_t ev_fork = nullptr, ev_join[MAX_LANES] = {}, ev_chain[MAX_LANES] = {};
    // Render-ahead for the one-iteration-per-call shape (ptx_iterate = the reference's pathtrace(iter)): lanes 1 and 2 take
    // turns tracing the NEXT kmax iterations into their per-iteration radiance buffers while the caller works the current
    // batch off, one k_gather (+ k_stats) per call on the main stream.  What a call returns is unchanged: the image holds
    // exactly the iterations asked for so far, summed in the same order.
    struct Ahead { int first = 0, count = 0, next = 0, unfolded = 0; bool use_cache = false, valid = false, timed = false;
                   unsigned long long dir_bins = ~0ull, ntab_bins = 0ull; };      // (the record masks the batch was written with: k_stats)
    Ahead ahead[MAX_LANES];
    int ahead_cur = -1, ahead_nxt = -1;                  // lane whose batch is being consumed / lane holding the batch after it
    bool render_ahead = false;
    int last_ahead_lane = -1;                            // != -1: the previous operation was a call served from that lane's batch
    hipEvent_t ev_ahead0[MAX_LANES] = {}, ev_ahead1[MAX_LANES] = {};
    int uses_uv = 0;
    unsigned long long dir_bins = ~0ull;                 // BounceParams::dir_bins (all ones: every record carries its direction)
    unsigned long long ntab_bins = 0ull;                 // BounceParams::ntab_bins (none: every record carries its normal)
    unsigned long long cache_dir_bins = ~0ull, cache_ntab_bins = 0ull;      // ... as the cached camera bounce was written
    int cache_idx16 = 0;                                                    // ... and its local index (BounceParams::idx16)
    uchar4 *d_pbo = nullptr;                             // ptx_write_pbo's device staging (allocated on first use)
    float *d_denoised = nullptr;                         // ptx_write_denoised_pbo_device's copy of the host frame (first use)
    float *d_albedo = nullptr;                           // apps variant only: W*H*3
    unsigned long long *d_stamps = nullptr;              // diagnostic build only
    float *d_part = nullptr;                             // [kmax][W*H*3] per-iteration radiance (batched mode)
    int32_t *d_cache_totals = nullptr;                   // [2][nbins] of bounce 0 (cache)
    int32_t *d_emit_count = nullptr, *d_emit_pix = nullptr; float *d_emit_rgb = nullptr;
    int64_t *d_stats = nullptr;                          // [64] last iteration, [64] = running total, [65] = fenced indices (BounceParams::fenced), [66..68] = stored paths: all, with direction, with normal code
    uint32_t fence_slots = 0;                            // = cap; PTX_DEBUG_FENCE_SLOTS lowers it (test of the counter: entries beyond it are fenced)
    int maxBounces = 0;
    bool cache_valid = false;
    int64_t iterations = 0;
    double loop_ms_total = 0.0;
    // optional per-kernel timing (bench.py's roofline leg): events around every launch of an iteration
    bool ktiming = false;
    std::vector<hipEvent_t> kev;                         // pairs (start, stop)
    std::vector<int> kev_kind;                           // per pair: 0 k_bounce<first>, 1 k_bounce, 2 k_mesh + k_finish, 3 pass 2 of the split bounce (the ranking pass)
    size_t kev_used = 0;
    // debug capture
    int capture_bounce = -1;
    int32_t *d_cap = nullptr;                            // pix, idx, mg [cap each] + totals
    float *d_cap_f = nullptr;                            // the 15 float fields [cap each]
    bool cap_filled = false;
    DScene scene() const {
        DScene s; s.geoms = d_geoms; s.mats = d_mats; s.faces = d_faces; s.tri9 = d_tri9; s.texels = d_texels; s.ngeoms = ngeoms; s.nmats = nmats;
        s.gtab = d_gtab; s.aabb = d_aabb; s.cull = 0; s.cube_bits = cube_bits; s.sphere_bits = sphere_bits; s.mesh_bits = mesh_bits;
        s.bvh_nodes = d_bvh_nodes; s.bvh_tris = d_bvh_tris; s.bvh_root = d_bvh_root; s.bvh_depth = d_bvh_depth; s.bvh_wide = d_bvh_wide; s.bvh_wroot = d_bvh_wroot; s.bvh_wneed = d_bvh_wneed; s.bvh_stack = 0; s.ntri_lds = 0; s.mesh_chunks = mesh_chunks;
        s.fnorm = d_fnorm; s.cnorm = d_cnorm; s.bump_bits = bump_bits;
        s.tri_lds = 0; s.ntri = ntri;      // tri_lds is switched on only by launches that stage the table (k_bounce)
        s.ldsblob = nullptr;               // (set by enqueue_batch together with tri_lds / ntri_lds: the blob is laid out for those)
        return s;
    }
    bool cache_active() const { return opt.cache_first_bounce && !opt.antialiasing && !opt.depth_of_field; }
    const KernelSet *ks = &g_kernels_here;               // the code object the arithmetic-bearing kernels are launched from (ptx_options.arith)
};

namespace {

// field arrays of `stride` elements each (stride = segments x cap: segment s of a field starts at s*cap)
void carve(PathSoA &s, float *f, int32_t *i, size_t stride) {
    s.q = f; s.i = i; s.stride = (uint32_t)stride;
}

PathSoA soa_shift(PathSoA s, size_t off) {       // host side of soa_offset: the same fields `off` elements further
    s.q += 4 * off; s.i += off;
    return s;
}

// multiplier and shift of fastdiv (device) for the divisor d >= 1
void fastdiv_magic(uint32_t d, uint32_t &mul, uint32_t &sh) {
    if (d <= 1) { mul = 0; sh = 255; return; }
    int l = 0;
    while ((1ull << l) < d) l++;
    sh = (uint32_t)(l - 1);
    mul = (uint32_t)(((1ull << (31 + l)) / d) + 1);
}

void camera_to_device(const ptx_camera &c, DCamera &d) {
    d.resx = c.resolution[0]; d.resy = c.resolution[1];
    memcpy(d.position, c.position, 12); memcpy(d.lookAt, c.lookAt, 12); memcpy(d.view, c.view, 12);
    memcpy(d.up, c.up, 12); memcpy(d.right, c.right, 12); memcpy(d.fov, c.fov, 8); memcpy(d.pixelLength, c.pixelLength, 8);
}

// Conservative world-space box of a geom for the candidate pre-test of intersectSceneCull: the transformed unit cube
// (which also contains the radius-0.5 sphere) or the transformed mesh vertices, evaluated in double and inflated by
// 1e-3 + 1e-4 * |coordinate| -- three orders of magnitude more than the fp32 error of the exact tests or of the slab
// pre-test itself, so a ray the exact test would report as a hit always reaches the box.  Anything non-finite gives an
// unbounded box (never culled).
void make_world_aabb(const DGeom &d, const std::vector<float> &faces, float out6[6]) {
    double lo[3] = {1e300, 1e300, 1e300}, hi[3] = {-1e300, -1e300, -1e300};
    bool ok = true;
    auto add = [&](double x, double y, double z) {
        for (int r = 0; r < 3; r++) {
            double v = (double)d.xf[0 * 4 + r] * x + (double)d.xf[1 * 4 + r] * y + (double)d.xf[2 * 4 + r] * z + (double)d.xf[3 * 4 + r];
            if (!(v == v) || v > 1e30 || v < -1e30) ok = false;
            lo[r] = std::min(lo[r], v); hi[r] = std::max(hi[r], v);
        }
    };
    if (d.type == G_OBJ) {
        if (d.faceCount == 0) ok = false;
        for (int j = 0; j < d.faceCount; j++)
            for (int k = 0; k < 3; k++) {
                const float *p = &faces[((size_t)d.faceStart + j) * 15 + k * 5];
                add(p[0], p[1], p[2]);
            }
    } else {
        for (int c = 0; c < 8; c++) add((c & 1) ? 0.5 : -0.5, (c & 2) ? 0.5 : -0.5, (c & 4) ? 0.5 : -0.5);
    }
    for (int r = 0; r < 3; r++) {
        if (!ok) { out6[r] = -INFINITY; out6[3 + r] = INFINITY; continue; }
        double m = 1e-3 + 1e-4 * std::max(std::fabs(lo[r]), std::fabs(hi[r]));
        out6[r] = nextafterf((float)(lo[r] - m), -INFINITY);
        out6[3 + r] = nextafterf((float)(hi[r] + m), INFINITY);
    }
}

// Which material bins' records must carry the incoming direction to the next bounce -- scatterRay (pt_device.h) reads it in its
// reflective and refractive branches and for every hit on an OBJ geom (Schlick's cosine), never for a diffuse cube or sphere hit --
// and which bins' hits are all cube hits (the material is on cubes only): their records carry a code for the cube's tabulated normal
// instead of the normal.  The next bounce tells the kinds of record apart by sorted position, at most two ranges each, so a mask keeps
// at most two runs of set bits: gaps between the runs of the first are FILLED (a direction more is harmless), runs beyond the second
// of the other are CLEARED.  nmaterials <= 64 bins (bin = nmaterials - 1 - material when sorting, else one bin).
void record_masks(int nmaterials, const DMaterial *mats, int ngeoms, const int *geom_type, const int *geom_material, bool sort,
                  unsigned long long &dir_bins, unsigned long long &ntab_bins) {
    unsigned long long need = 0, cubes = 0;
    const int nbins = sort ? nmaterials : 1;
    for (int m = 0; m < nmaterials; m++) {
        bool nd = mats[m].hasReflective > 0 || mats[m].hasRefractive > 0, on_cube = false, on_other = false;
        for (int i = 0; i < ngeoms; i++)
            if (geom_material[i] == m) { nd = nd || geom_type[i] == G_OBJ; (geom_type[i] == G_CUBE ? on_cube : on_other) = true; }
        const int b = sort ? nmaterials - 1 - m : 0;
        if (nd) need |= 1ull << b;
        if (on_cube && !on_other && sort) cubes |= 1ull << b;
    }
    auto runs = [&](unsigned long long mask) { int n = 0; for (int b = 0; b < nbins; b++) n += ((mask >> b) & 1) && !(b && ((mask >> (b - 1)) & 1)); return n; };
    while (runs(need) > 2) {                          // fill the gap behind the first run
        int b = 0;
        while (!((need >> b) & 1)) b++;
        while ((need >> b) & 1) b++;
        need |= 1ull << b;
    }
    while (runs(cubes) > 2) cubes &= ~(1ull << (63 - __builtin_clzll(cubes)));      // drop the highest set bin
    dir_bins = need; ntab_bins = cubes;
}

// The same box as the device's candidate pre-test reads it (cullMask): centre and half extent.  The half extent grows by what the two
// roundings can lose (half an ulp of the centre, half an ulp of itself) and then some; an unbounded box is centre 0, half extent inf.
void world_box_centre_half(const float lohi8[8], float out8[8]) {
    for (int r = 0; r < 3; r++) {
        const double lo = lohi8[r], hi = lohi8[4 + r];
        if (!(lo > -1e30 && hi < 1e30)) { out8[r] = 0.f; out8[4 + r] = INFINITY; continue; }
        const float c = (float)(0.5 * (lo + hi));
        const double h = std::max(hi - (double)c, (double)c - lo);
        out8[r] = c;
        out8[4 + r] = nextafterf((float)(h + 2e-7 * (std::fabs((double)c) + h)), INFINITY);
    }
    out8[3] = out8[7] = 0.f;
}

// Which geoms can the camera rays of a tile reach at all?  Per geom the pixel rectangle that its conservative world box projects
// into (double precision, widened by the antialiasing jitter and two more pixels); a corner at or behind the eye plane makes it the
// whole frame.  A tile is 256 consecutive OWNED pixels: one span of a row, or -- when it wraps -- whole rows.  Bit g of a tile's
// word is cleared only when geom g's rectangle misses the tile's: a superset of what any of its rays can hit, so the candidate
// masks built from it (cullMask<SUBSET>) hold exactly the bits the full loop would set for those rays.  Recomputed when the camera
// changes; not where the candidate masks are off.
// With depth of field (generateRay: src/pathtrace.cu:236-251) a pixel's rays leave a lens -- the eye moved by up to lensRadius 0.8 in
// WORLD x / y -- towards the pixel's focus point pFocus on the plane z = eye.z +- 11.  A point c is on such a ray iff
// c = e + mu (pFocus - e), mu = (c.z - eye.z) / (+-11) > 0, i.e. pFocus = centre(c) + l (1 - 1/mu) with centre(c) the perspective
// image of c on the focus plane and |l| <= 0.8: a disc of radius rho(c) = 0.8 |1 - 1/mu| around it.  Over a box, centre() is a
// projective map (the hull of the corners' images) and rho is extremal at a corner, so the box's focus points lie within the corners'
// images widened by the LARGEST corner radius; each is then projected to pixels through the pinhole (the relation between a pixel and
// its focus point) at the four corners of its bounding square.  Anything doubtful -- a corner not in front of the lens plane, a frame
// whose pixels do not all look towards the same side of it -- keeps the whole frame.
// (pure host arithmetic: ptx_debug_tile_geoms hands it to the CPU tests, which check the superset property ray by ray)
void tile_geom_masks(const DCamera &c, const TileMap &tm, int maxTiles, int ngeoms, const float *aabb8, bool dof, std::vector<uint32_t> &masks) {
    const int W = c.resx, H = c.resy;
    // p - eye = l * (view - R sx - U sy),  R = right * pixelLength.x, U = up * pixelLength.y,  sx = x - W/2, sy = y - H/2  (generateRay)
    const double V[3] = {c.view[0], c.view[1], c.view[2]};
    const double R[3] = {(double)c.right[0] * c.pixelLength[0], (double)c.right[1] * c.pixelLength[0], (double)c.right[2] * c.pixelLength[0]};
    const double U[3] = {(double)c.up[0] * c.pixelLength[1], (double)c.up[1] * c.pixelLength[1], (double)c.up[2] * c.pixelLength[1]};
    // solve [V  -R  -U] (l, l sx, l sy)^T = p - eye by Cramer's rule
    auto det3 = [](const double *a, const double *b, const double *d) {
        return a[0] * (b[1] * d[2] - b[2] * d[1]) - b[0] * (a[1] * d[2] - a[2] * d[1]) + d[0] * (a[1] * b[2] - a[2] * b[1]);
    };
    const double nR[3] = {-R[0], -R[1], -R[2]}, nU[3] = {-U[0], -U[1], -U[2]};
    const double D = det3(V, nR, nU);
    // pixel coordinates of the point eye + p; false: not safely in front of the eye
    auto project = [&](const double p[3], double &x, double &y) {
        const double l = det3(p, nR, nU) / D, lsx = det3(V, p, nU) / D, lsy = det3(V, nR, p) / D;
        // in front of the eye by a margin relative to the point's distance (|view| = 1): otherwise the projection is meaningless
        if (!(l > 1e-6 * (std::fabs(p[0]) + std::fabs(p[1]) + std::fabs(p[2])) && l > 1e-12)) return false;
        x = lsx / l + W * 0.5; y = lsy / l + H * 0.5;
        return std::isfinite(x) && std::isfinite(y);
    };
    double zsign = 0.0;                                      // depth of field: the side of the lens plane every pixel looks to
    bool dof_ok = true;
    if (dof) {
        for (int k = 0; k < 4; k++) {                        // the frame's corner pixels (+- the jitter): z of the unnormalised direction
            const double sx = ((k & 1) ? W + 1.0 : -1.0) - W * 0.5, sy = ((k & 2) ? H + 1.0 : -1.0) - H * 0.5;
            const double dx = V[0] - R[0] * sx - U[0] * sy, dy = V[1] - R[1] * sx - U[1] * sy, dz = V[2] - R[2] * sx - U[2] * sy;
            const double z = dz / std::sqrt(dx * dx + dy * dy + dz * dz);
            if (!(std::fabs(z) > 0.05) || (zsign != 0.0 && (z > 0) != (zsign > 0))) dof_ok = false;
            zsign = z > 0 ? 1.0 : -1.0;
        }
    }
    std::vector<int> rect((size_t)ngeoms * 4);
    for (int g = 0; g < ngeoms; g++) {
        int *r = &rect[(size_t)g * 4];
        r[0] = 0; r[1] = W - 1; r[2] = 0; r[3] = H - 1;                  // x0, x1, y0, y1: the whole frame unless proven smaller
        const float *b = aabb8 + (size_t)g * 8;
        bool ok = std::isfinite(D) && std::fabs(D) > 1e-30 && dof_ok;
        double xlo = 1e300, xhi = -1e300, ylo = 1e300, yhi = -1e300;
        double cp[8][3], rho = 0.0;
        for (int k = 0; k < 8 && ok; k++) {
            cp[k][0] = (double)((k & 1) ? b[4] : b[0]) - c.position[0]; cp[k][1] = (double)((k & 2) ? b[5] : b[1]) - c.position[1];
            cp[k][2] = (double)((k & 4) ? b[6] : b[2]) - c.position[2];
            if (!std::isfinite(cp[k][0]) || !std::isfinite(cp[k][1]) || !std::isfinite(cp[k][2])) ok = false;
            if (ok && dof) {
                const double mu = cp[k][2] / (zsign * 11.0);             // focalDistance 11 (src/pathtrace.cu:238)
                if (!(mu > 1e-3)) { ok = false; break; }                 // not in front of the lens plane: no bound from this corner
                rho = std::max(rho, 0.8 * std::fabs(1.0 - 1.0 / mu) * 1.0001 + 1e-6);      // lensRadius .8 (:237)
            }
        }
        for (int k = 0; k < 8 && ok; k++) {
            if (!dof) {
                double x, y;
                if (!project(cp[k], x, y)) { ok = false; break; }
                xlo = std::min(xlo, x); xhi = std::max(xhi, x); ylo = std::min(ylo, y); yhi = std::max(yhi, y);
                continue;
            }
            const double mu = cp[k][2] / (zsign * 11.0);
            const double fx = cp[k][0] / mu, fy = cp[k][1] / mu;         // the corner's image on the focus plane (relative to the eye)
            for (int q = 0; q < 4 && ok; q++) {
                const double p[3] = {fx + ((q & 1) ? rho : -rho), fy + ((q & 2) ? rho : -rho), zsign * 11.0};
                double x, y;
                if (!project(p, x, y)) { ok = false; break; }
                xlo = std::min(xlo, x); xhi = std::max(xhi, x); ylo = std::min(ylo, y); yhi = std::max(yhi, y);
            }
        }
        if (!ok || !(xlo <= xhi) || !(ylo <= yhi)) continue;
        // a pixel's rays cover [x - 0.5, x + 0.5] (antialiasing jitter, generateRay); two more pixels for the fp32 ray arithmetic
        const double m = 2.5;
        r[0] = (int)std::max(0.0, std::min((double)W, std::floor(xlo - m)));
        r[1] = (int)std::max(-1.0, std::min((double)W - 1, std::ceil(xhi + m)));
        r[2] = (int)std::max(0.0, std::min((double)H, std::floor(ylo - m)));
        r[3] = (int)std::max(-1.0, std::min((double)H - 1, std::ceil(yhi + m)));
    }
    masks.assign((size_t)maxTiles, 0u);
    auto owned_xy = [&](int i, int &x, int &y) {                        // = owned_pixel (device)
        const int r = i / W;
        x = i - r * W;
        if (tm.tile_world <= 1) { y = r; return; }
        const int k = r / tm.tile_rows;
        y = (k * tm.tile_world + tm.tile_rank) * tm.tile_rows + (r - k * tm.tile_rows);
    };
    for (int tile = 0; tile < maxTiles; tile++) {
        const int i0 = tile * TILE, i1 = std::min(i0 + TILE, tm.owned) - 1;
        if (i1 < i0) { masks[tile] = 0xffffffffu; continue; }
        int x0, y0, x1, y1;
        owned_xy(i0, x0, y0); owned_xy(i1, x1, y1);
        if (y0 != y1) { x0 = 0; x1 = W - 1; }                            // wraps: whole rows y0 .. y1 (rows of other ranks in between included)
        uint32_t m = 0;
        for (int g = 0; g < ngeoms; g++) {
            const int *r = &rect[(size_t)g * 4];
            if (!(r[1] < x0 || r[0] > x1 || r[3] < y0 || r[2] > y1)) m |= 1u << g;
        }
        masks[tile] = m;
    }
}

int update_tile_geoms(ptx_tracer *t) {
    t->tile_geoms_valid = false;
    if (!t->cull || t->ngeoms > 32 || t->ngeoms < 1 || getenv("PTX_DEBUG_NO_TILE_GEOMS")) return PTX_OK;
    const bool dof = t->opt.depth_of_field != 0;
    if (dof && getenv("PTX_DEBUG_NO_TILE_GEOMS_DOF")) return PTX_OK;
    std::vector<uint32_t> masks;
    tile_geom_masks(t->cam, t->tm, t->maxTiles, t->ngeoms, t->h_aabb.data(), dof, masks);
    HIPCHECK(hipMemcpyAsync(t->d_tile_geoms, masks.data(), sizeof(uint32_t) * masks.size(), hipMemcpyHostToDevice, t->stream));
    HIPCHECK(hipStreamSynchronize(t->stream));      // (masks is a local)
    t->tile_geoms_valid = true;
    return PTX_OK;
}

int free_tracer(ptx_tracer *t) {
    if (!t) return PTX_OK;
    hipSetDevice(t->device);
    if (t->stream) hipStreamSynchronize(t->stream);
    for (int l = 1; l < MAX_LANES; l++) if (t->lane_stream[l]) hipStreamSynchronize(t->lane_stream[l]);      // work traced ahead
    hipFree(t->d_geoms); hipFree(t->d_mats); hipFree(t->d_faces); hipFree(t->d_tri9); hipFree(t->d_gtab); hipFree(t->d_aabb); hipFree(t->d_tile_geoms); hipFree(t->d_bvh_nodes); hipFree(t->d_bvh_tris); hipFree(t->d_bvh_root); hipFree(t->d_bvh_depth); hipFree(t->d_bvh_wide); hipFree(t->d_bvh_wroot); hipFree(t->d_bvh_wneed); hipFree(t->d_keys); hipFree(t->d_tile_done); hipFree(t->d_items); hipFree(t->d_item_count); hipFree(t->d_fnorm); hipFree(t->d_cnorm); hipFree(t->d_ldsblob); hipFree(t->d_texels);
    if (t->own_image) hipFree(t->d_image);
    for (int k = 0; k < 3; k++) { hipFree(t->d_fbuf[k]); hipFree(t->d_ibuf[k]); }
    hipFree(t->d_cache_chunk); hipFree(t->d_cache_super);
    hipFree(t->d_counts); hipFree(t->d_chunk); hipFree(t->d_totals); hipFree(t->d_cache_totals);
    hipFree(t->d_emit_count); hipFree(t->d_emit_pix); hipFree(t->d_emit_rgb); hipFree(t->d_stats); hipFree(t->d_cap); hipFree(t->d_cap_f); hipFree(t->d_part); hipFree(t->d_albedo); hipFree(t->d_stamps); hipFree(t->d_pbo); hipFree(t->d_denoised);
    for (hipEvent_t e : t->kev) hipEventDestroy(e);
    if (t->ev_start) hipEventDestroy(t->ev_start);
    if (t->ev_stop) hipEventDestroy(t->ev_stop);
    for (int l = 1; l < MAX_LANES; l++) if (t->lane_stream[l]) hipStreamDestroy(t->lane_stream[l]);
    if (t->ev_fork) hipEventDestroy(t->ev_fork);
    for (hipEvent_t e : t->ev_join) if (e) hipEventDestroy(e);
    for (hipEvent_t e : t->ev_chain) if (e) hipEventDestroy(e);
    for (hipEvent_t e : t->ev_ahead0) if (e) hipEventDestroy(e);
    for (hipEvent_t e : t->ev_ahead1) if (e) hipEventDestroy(e);
    if (t->own_stream && t->stream) hipStreamDestroy(t->stream);
    delete t;
    return PTX_OK;
}

// The specialised variants k_bounce<., ., FAST> hard-wire option values inside the kernel (sort = 1, no cache fill, no albedo, per-
// iteration radiance buffers; the unsplit one also: all scene tables in LDS, no texture, no bump map, no BVH, no depth of field).
// Run outside them they index tables the host sized for OTHER values: with sort_by_material = 0 the host has ONE bin, the kernel
// computes bin = nmats - 1 - material and writes past every per-bin table in LDS and in global memory -- that was round 2's fault
// (gpurun_out/quick_S1.log: an A/B build that took the variant unconditionally; wrong frames on the cache-filling pass and with
// depth of field, then "Memory access fault" on cornellObj with sort_by_material = 0).  So there is ONE predicate, used by every
// launch site: it names the first assumption that does not hold (nullptr: all hold), and launch_bounce refuses -- PTX_ERR_INVALID,
// nothing is launched -- when the variant is asked for regardless (PTX_DEBUG_FORCE_FAST, tests only).
const char *fast_violation(const ptx_tracer *t, int mode, bool first, bool needs_albedo, const BounceParams &bp) {
    if (!bp.part) return "no per-iteration radiance buffers (part == NULL): the variant stores, it never adds to the image";
    if (!bp.sort || bp.nbins != std::max(t->nmats, 1)) return "sort_by_material = 0: the per-bin tables hold one bin";
    if (bp.nbins > 64) return "more than 64 material bins (lane b of a wave owns bin b in the variant's tile epilogue)";
    if (bp.emit_count) return "the cache-filling pass (emit_count != NULL) records bounce-0 light hits";
    if (needs_albedo) return "the launch set contains iteration 1 of the apps variant (albedo AOV)";
    if (!bp.sc.cull || !bp.sc.tri_lds) return "candidate masks or LDS scene tables are off";
    if (mode == 0) {
        if (t->split_mesh) return "the scene takes the split mesh search";
        if (first && bp.dof) return "depth of field on the camera-ray bounce";
        if (bp.uses_uv) return "textured scene";
        if (bp.sc.bump_bits) return "bump-mapped mesh";
        if (bp.sc.ntri_lds != bp.sc.ntri) return "triangle tables not staged in LDS";
        if (bp.sc.bvh_root) return "mesh with a BVH";
    } else if (!t->split_mesh || !bp.keys || !bp.items || !bp.item_count) return "not the split mesh search";
    return nullptr;
}

// the one launch site of k_bounce: picks the variant by the predicate above
int launch_bounce(const ptx_tracer *t, bool first, int mode, bool needs_albedo, dim3 grid, size_t lds, hipStream_t stream, const BounceParams &bp) {
    const char *why = fast_violation(t, mode, first, needs_albedo, bp);
    bool fast = !t->no_fast && !why;
    if (t->force_fast) {
        if (why) return set_error(PTX_ERR_INVALID, std::string("specialised k_bounce requested outside its preconditions: ") + why);
        fast = true;
    }
    t->ks->bounce(first ? 1 : 0, mode, fast ? 1 : 0, grid, lds, stream, &bp);
    return PTX_OK;
}

// Enqueues K iterations (iter_first, iter_first + stride, ...) as K segments of every launch: blockIdx.y picks
// the segment, each segment is an independent stream with its own buffers, so the launches carry K times the work
// (what keeps a 1/8-frame tile of a multi-GPU run, or the thin late bounces, from being launch- and tail-bound).
// Lane `lane` works on segments lane*kmax .. of every per-iteration buffer and on its own stream; the image is touched
// only by k_gather, and the gathers of successive batches are chained by events (prev_lane = the lane the previous batch
// ran on), so the fp32 sums happen in iteration order whatever the overlap.
int enqueue_batch(ptx_tracer *t, int iter_first, int K, int stride = 1, int lane = 0, int prev_lane = -1, bool defer = false) {
    hipStream_t stream = lane == 0 ? t->stream : t->lane_stream[lane];
    const size_t seg0 = (size_t)lane * t->kmax;
    const int nb = t->nbins;
    const int ntri_lds = t->split_mesh ? 0 : t->ntri_lds;
    const int triWords = t->tri_lds ? sceneTableWords(ntri_lds, t->nmats, t->ngeoms) : 0;
    const size_t lds_bounce = sizeof(int32_t) * (bounceLdsWords(triWords, nb) + (t->split_mesh ? QUEUE_WORDS : 0)) + (size_t)t->dbg_extra_lds;      // (+ PTX_DEBUG_EXTRA_LDS bytes: occupancy experiments)
    const bool cache_on = t->cache_active();
    const bool use_cache = cache_on && t->cache_valid && iter_first != 1;
    const bool fill_cache = cache_on && !use_cache;
    const bool batched = K > 1 || t->lanes > 1;      // ending paths store into per-iteration buffers, k_gather sums them
    // the specialised unsplit kernel runs 5 workgroups per CU at a time: a grid of 8 per CU would be 1.6 rounds of them (C4 -1.3 %)
    // (the apps variant's x PI at the deposit stays a run-time value in every kernel; its albedo AOV is written by iteration 1 alone,
    // so only a launch set that contains iteration 1 needs the general kernel for it)
    const bool needs_albedo = t->d_albedo && iter_first == 1;
    // (only the grid size depends on this estimate; what is launched is decided per launch by fast_violation)
    const bool fast_unsplit = !t->split_mesh && !t->no_fast && batched && !t->uses_uv && t->opt.sort_by_material &&
                              !needs_albedo && t->cull && t->tri_lds && t->bump_bits == 0 && t->ntri_lds == t->ntri &&
                              !t->d_bvh_root;
    // ... the split bounce's kernels are many short ones: 16 workgroups per CU (C5 -2 %); everything else 8 as before
    // (traced ahead of per-call requests: two of the seven slots per CU stay free, so that the caller's own short kernels -- gather,
    // preview -- start at once instead of waiting for one of these long-running workgroups to end: 0.53 -> 0.50 ms per call)
    // The launch's workgroups: one round of the kernel's occupancy over the whole chip, WHATEVER one segment holds (round 3: until
    // then the total was capped by one segment's tile count, so the K = 10 segments of a 1/8 tile ran as 1010 workgroups of ten
    // tiles -- 4 per CU -- instead of 1790 of six: 20 steps of such a tile 0.636 -> 0.585 ms, tools/gpu_tile_grid_sweep.py)
    // Round 4: MORE workgroups than one round of the occupancy.  A workgroup owns a contiguous chunk of tiles, and chunks are unequal --
    // the camera-ray bounce's tiles cost anything from nothing (a tile that sees no geom) to a full tile, the later bounces' tiles
    // differ by material mix -- so with one chunk per resident workgroup a kernel ends when its heaviest chunk does; and the other launch
    // sets' short kernels (k_finish, the ranking pass) get a slot only when a workgroup of the long one retires.  Measured on one box
    // each (gpurun_out/r4_c4wg*.log, r4_bouncewg*.log): C4 camera bounce alone 0.0365 (7 per CU) / 0.0338 (14) / 0.0320 (21) / 0.0300 ms
    // (28), later bounces 0.163 / 0.161 / 0.160 / 0.160 / 0.173 (42), wall of the 20-step run 0.176 / 0.172 / 0.173 / 0.173 / 0.179;
    // C5 (split bounce) 16 per CU 1.06-1.10, 32 1.03-1.07, 48 1.04-1.07 ms per iteration.
    // End of round 4, after the records' diet (32 B instead of 56 for a wall hit, 16-byte quads) and with eight workgroups' LDS per CU: the
    // later bounces as ONE round of the occupancy again (8 per CU) and the camera bounce 20 -- C4's 20-step run 0.147 -> 0.1425 ms per
    // step, its long run 0.1415 -> 0.138 (three runs each of six plans on one box, gpurun_out/r4grid2.log; 28 / 14 was the choice while
    // the kernels moved 40 % more bytes and a chunk's tail of index traffic was worth spreading).
    // (traced ahead of per-call requests: five per CU, so that two slots stay free for the caller's own short kernels)
    auto per_cu = [&](bool first_bounce) {
        if (fast_unsplit && !defer) { if (const char *e = getenv(first_bounce ? "PTX_DEBUG_WG_FIRST" : "PTX_DEBUG_WG_LATER")) return std::max(1, atoi(e)); }      // tuning experiments
        if (fast_unsplit) return defer ? PT_FAST_WAVES - 2 : first_bounce ? 20 : PT_FAST_WAVES;
        return t->split_mesh ? 32 : 8;
    };
    auto gx_of = [&](bool first_bounce) {
        int grid = t->grid_forced ? t->grid : t->cus * per_cu(first_bounce);
        if (t->dbg_total_wg_per_cu > 0) grid = t->cus * t->dbg_total_wg_per_cu;      // tuning experiments
        int g = grid / K;                            // workgroups per segment
        if (!t->grid_forced && t->dbg_total_wg_per_cu <= 0) {
            // ... but only where a workgroup keeps at least ~8 tiles of the camera bounce: a rank's 1/8 tile is a chain of dependent round
            // trips per workgroup life, and more workgroups are more prologues there -- 20 steps of such a tile 0.54 -> 0.60 ms
            // with the larger grids.  Below that the grid is one round of the occupancy, as before.
            const int base = t->cus * (fast_unsplit ? (defer ? PT_FAST_WAVES - 2 : PT_FAST_WAVES) : t->split_mesh ? 16 : 8) / K;
            g = std::min(g, std::max(base, t->maxTiles / 8));
        }
        if (g < 64 && t->dbg_total_wg_per_cu <= 0) g = 64;
        if (g > t->maxTiles) g = t->maxTiles;
        if (g > t->grid_seg) g = t->grid_seg;
        if (g > grid) g = grid;
        return g < 1 ? 1 : g;
    };
    const int gx_first = gx_of(true), gx_later = gx_of(false);
    const int grid = t->cus * (t->dbg_total_wg_per_cu > 0 ? t->dbg_total_wg_per_cu : per_cu(false));      // (k_finish's grid-stride launch)
    const int gx = gx_later;
    const int nsuper = (gx + 63) / 64;
    const size_t chunk_cap = (size_t)nb * t->grid_seg;                 // runs per table at most
    const size_t seg_counts = (2 * (size_t)nb + 1) * t->maxTiles, seg_chunk = 2 * 3 * chunk_cap, seg_totals = t->seg_totals;      // (two prefix tables + stored paths per tile)
    int32_t *counts_all = t->d_counts + seg0 * seg_counts, *counts_scat = counts_all + (size_t)nb * t->maxTiles;
    // run tables of bounce b: parity b & 1 of the segment's pair (bounce b + 1 reads them while it writes its own)
    auto chunks = [&](int bounce) { return t->d_chunk + seg0 * seg_chunk + (size_t)(bounce & 1) * 3 * chunk_cap; };
    auto totals = [&](int bounce, int which) { return t->d_totals + seg0 * seg_totals + ((size_t)bounce * 2 + which) * nb; };
    auto supers = [&](int bounce, int which) { return t->d_super + seg0 * seg_totals + ((size_t)bounce * 2 + which) * nb * t->nsuper; };
    // per-bounce totals and group totals are accumulated with atomics: clear them once per batch
    HIPCHECK(hipMemsetAsync(t->d_totals + seg0 * seg_totals, 0, sizeof(int32_t) * seg_totals * (size_t)K, stream));
    // ... and the batch's "lit" flags (behind each segment's radiance: cap bytes per segment, one strided memset)
    if (K > 1 || t->lanes > 1)
        HIPCHECK(hipMemset2DAsync(reinterpret_cast<char *>(t->d_part + seg0 * t->seg_part) + sizeof(float) * 3 * (size_t)t->cap, sizeof(float) * t->seg_part, 0,
                                  (size_t)t->cap, (size_t)K, stream));

    // per-kernel timing brackets (only when switched on; costs two event records per launch)
    auto kt_begin = [&](int kind) -> int {
        if (!t->ktiming) return PTX_OK;
        if (t->kev_used + 2 > t->kev.size()) {
            hipEvent_t a, b2;
            HIPCHECK(hipEventCreate(&a)); HIPCHECK(hipEventCreate(&b2));
            t->kev.push_back(a); t->kev.push_back(b2);
        }
        if (t->kev_kind.size() < t->kev.size() / 2) t->kev_kind.resize(t->kev.size() / 2);
        t->kev_kind[t->kev_used / 2] = kind;
        HIPCHECK(hipEventRecord(t->kev[t->kev_used], stream));
        return PTX_OK;
    };
    auto kt_end = [&]() -> int {
        if (!t->ktiming) return PTX_OK;
        HIPCHECK(hipEventRecord(t->kev[t->kev_used + 1], stream));
        t->kev_used += 2;
        return PTX_OK;
    };
#define KT(kind, launch) do { int rc_ = kt_begin(kind); if (rc_ != PTX_OK) return rc_; launch; rc_ = kt_end(); if (rc_ != PTX_OK) return rc_; } while (0)
    if (fill_cache) HIPCHECK(hipMemsetAsync(t->d_emit_count, 0, sizeof(int32_t), stream));
    // what this batch's launches WRITE their records with (a debug capture switches the masks off for its own launches); k_stats weighs
    // the batch's stored paths by these, not by the tracer's
    const bool masks_off = t->capture_bounce >= 0 && !getenv("PTX_DEBUG_KEEP_DIR_SKIP");
    const unsigned long long batch_dir_bins = masks_off ? ~0ull : t->dir_bins, batch_ntab_bins = masks_off ? 0ull : t->ntab_bins;
    for (int b = 0; b < t->traceDepth; b++) {
        const bool first = b == 0;
        if (first && use_cache) {
            // first-bounce cache: the sorted bounce-0 stream and its light hits are identical every iteration
            // when primary rays are not jittered, so bounce 0 is skipped (intent of src/pathtrace.cu:492-499,514)
            hipLaunchKernelGGL(k_seed_totals, dim3(1), dim3(256), 0, stream, totals(0, 0), seg_totals, K, t->d_cache_totals, 2 * nb);
            if (batched) {
                // (the cached bounce-0 misses need nothing: their flags were cleared with the batch's)
                const size_t seg_part = t->seg_part;
                hipLaunchKernelGGL(k_replay_emission, dim3(64), dim3(256), 0, stream, t->tm, t->d_emit_count, t->d_emit_pix, t->d_emit_rgb,
                                   t->d_part + seg0 * seg_part, seg_part, K, 0, (uint32_t)t->cap);
            } else {
                hipLaunchKernelGGL(k_replay_emission, dim3(64), dim3(256), 0, stream, t->tm, t->d_emit_count, t->d_emit_pix, t->d_emit_rgb,
                                   t->d_image, (size_t)0, 1, 1, (uint32_t)t->cap);
            }
            continue;
        }
        BounceParams bp;
        bp.sc = t->scene(); bp.sc.tri_lds = t->tri_lds; bp.sc.ntri_lds = t->split_mesh ? 0 : t->ntri_lds; bp.sc.cull = t->cull; bp.cam = t->cam; bp.tm = t->tm;
        bp.sc.ldsblob = t->d_ldsblob;
        // (split: the mesh search runs in k_mesh from global memory, so the triangle tables need no LDS)
        // bounce b writes its stage into soa[1 - (b & 1)] (bounce 0 of a cache-enabled tracer: into soa[2], kept across
        // iterations) and reads the previous bounce's through that bounce's local index and run tables
        const bool from_cache = (b == 1 && cache_on), to_cache = (first && cache_on);
        bp.in = from_cache ? t->soa[2] : soa_shift(t->soa[b & 1], seg0 * t->cap);
        bp.stage = to_cache ? t->soa[2] : soa_shift(t->soa[1 - (b & 1)], seg0 * t->cap);
        bp.in_totals = first ? nullptr : from_cache ? t->d_cache_totals : totals(b - 1, 0);
        bp.in_super = first ? nullptr : from_cache ? t->d_cache_super : supers(b - 1, 0);
        bp.in_chunk = first ? nullptr : from_cache ? t->d_cache_chunk : chunks(b - 1);
        const int gx_b = first ? gx_first : gx_later;                       // this launch's workgroups per segment
        bp.in_gx = from_cache ? t->cache_gx : (b == 1 ? gx_first : gx_later);      // ... and those of the launch whose run tables it reads
        bp.seg_in_totals = from_cache ? 0 : seg_totals; bp.seg_in_chunk = from_cache ? 0 : seg_chunk;
        bp.image = t->d_image;
        bp.iter = iter_first; bp.iter_stride = stride; bp.traceDepth = t->traceDepth; bp.bounce = b;
        bp.aa = t->opt.antialiasing; bp.dof = t->opt.depth_of_field; bp.sort = t->opt.sort_by_material; bp.uses_uv = t->uses_uv; bp.dir_bins = batch_dir_bins;
        bp.ntab_bins = batch_ntab_bins; bp.apps = t->opt.apps_variant; bp.albedo = t->d_albedo;
        // (the masks tell the READER of a stage what its records hold; a launch that WRITES with other masks than it reads with -- a cached
        // camera bounce replayed while a debug capture has switched them off, or the other way round -- gets both: in_* for what it reads)
        bp.in_dir_bins = from_cache ? t->cache_dir_bins : bp.dir_bins; bp.in_ntab_bins = from_cache ? t->cache_ntab_bins : bp.ntab_bins;
        // (one-word local index where no workgroup's chunk can pass 128 tiles: this launch's, and that of the launch whose stage it reads)
        auto idx16_of = [&](int gxw) { return (!getenv("PTX_DEBUG_NO_IDX16") && (t->maxTiles + gxw - 1) / std::max(gxw, 1) <= 128) ? 1 : 0; };
        bp.idx16 = idx16_of(first ? gx_first : gx_later);
        bp.in_idx16 = first ? 0 : from_cache ? t->cache_idx16 : idx16_of(b == 1 ? gx_first : gx_later);
        bp.nbins = nb; bp.maxTiles = t->maxTiles;
        bp.counts_all = counts_all; bp.counts_scat = counts_scat;
        bp.chunk = to_cache ? t->d_cache_chunk : chunks(b); bp.chunk_cap = (int32_t)chunk_cap;
        bp.super_all = supers(b, 0); bp.super_scat = supers(b, 1);
        bp.totals_all = totals(b, 0); bp.totals_scat = totals(b, 1);
        bp.nsuper = t->nsuper;
        bp.seg_in = from_cache ? 0 : (size_t)t->cap; bp.seg_stage = to_cache ? 0 : (size_t)t->cap;
        bp.seg_counts = seg_counts; bp.seg_chunk = to_cache ? 0 : seg_chunk; bp.seg_totals = seg_totals;
        bp.stamps = t->d_stamps;
        bp.seg_part = t->seg_part;
        bp.part = batched ? t->d_part + seg0 * bp.seg_part : nullptr;
        bp.emit_count = (first && fill_cache) ? t->d_emit_count : nullptr;
        bp.emit_pix = t->d_emit_pix; bp.emit_rgb = t->d_emit_rgb;
        bp.fenced = reinterpret_cast<unsigned long long *>(t->d_stats + 65); bp.fence_slots = t->fence_slots; bp.fence_slots_cap = (uint32_t)t->cap;
        bp.tile_geoms = (first && t->tile_geoms_valid) ? t->d_tile_geoms : nullptr;
        if (t->split_mesh) {
            bp.keys = t->d_keys + seg0 * (size_t)t->cap; bp.seg_keys = (size_t)t->cap;
            bp.items = t->d_items + seg0 * t->seg_items; bp.seg_items = t->seg_items;
            bp.item_count = t->d_item_count + 2 * seg0;          // (a launch set's K counts, then its K cursors: one memset)
            bp.item_cursor = bp.item_count + K;
            bp.tile_done = (first && t->d_tile_done) ? t->d_tile_done + seg0 * (size_t)t->maxTiles : nullptr;
            HIPCHECK(hipMemsetAsync(bp.item_count, 0, sizeof(int32_t) * 2 * (size_t)K, stream));
            KT(first ? 0 : 1, { int rcl = launch_bounce(t, first, 1, needs_albedo, dim3(gx_b, K), lds_bounce, stream, bp); if (rcl != PTX_OK) return rcl; });
            // the per-lane traversal stack lives in LDS and is what limits k_mesh's occupancy: as many entries as the longest walk needs
            // (k_mesh's waves draw from the segment's queue until it is empty: one round of the kernel's occupancy is all the grid needs)
            // Three workgroups per CU, not the five its LDS would admit: a workgroup holds 31 KB (the walks' stacks), five of them nearly all
            // of a CU's 160 KB -- while k_mesh runs, the OTHER launch sets' kernels then find no LDS to start in and the overlap of the three
            // sets stops.  With three, k_mesh alone is 4 % slower and C5's wall time 2 % shorter (1.066 -> 1.045 ms per iteration; 4: 1.054,
            // 2: 1.059; nine runs each on one box); a wave also draws from 460 rays instead of 270.
            const int mesh_gx = std::max(1, t->cus * (t->dbg_mesh_wg_per_cu > 0 ? t->dbg_mesh_wg_per_cu : PT_MESH_WG_PER_CU) / K);
            KT(2, { t->ks->mesh(first ? 1 : 0, dim3(mesh_gx, K), sizeof(int32_t) * ((size_t)t->bvh_stack * 256 + 32 * MESH_GEOM_WORDS), stream, &bp, t->bvh_stack);
                    t->ks->finish(first ? 1 : 0, dim3(std::max(1, grid / K), K), stream, &bp); });
            const size_t lds_pass2 = sizeof(int32_t) * ((size_t)ldsHeadWords(nb) + std::max<size_t>(TILE, 2 * (size_t)WAVES * nb));      // (ranking head + the second histogram; one key per slot in the three-barrier form)
            KT(3, { int rcl = launch_bounce(t, first, 2, needs_albedo, dim3(gx_b, K), lds_pass2, stream, bp); if (rcl != PTX_OK) return rcl; });
        } else {
            bp.keys = nullptr; bp.items = nullptr; bp.item_count = nullptr; bp.item_cursor = nullptr; bp.seg_keys = bp.seg_items = 0; bp.tile_done = nullptr;
            KT(first ? 0 : 1, { int rcl = launch_bounce(t, first, 0, needs_albedo, dim3(gx_b, K), lds_bounce, stream, bp); if (rcl != PTX_OK) return rcl; });
        }

        if (first && fill_cache) {
            HIPCHECK(hipMemcpyAsync(t->d_cache_totals, totals(0, 0), sizeof(int32_t) * 2 * nb, hipMemcpyDeviceToDevice, stream));
            HIPCHECK(hipMemcpyAsync(t->d_cache_super, supers(0, 0), sizeof(int32_t) * 2 * nb * t->nsuper, hipMemcpyDeviceToDevice, stream));
            t->cache_gx = gx_b;
            t->cache_dir_bins = bp.dir_bins; t->cache_ntab_bins = bp.ntab_bins; t->cache_idx16 = bp.idx16;
            t->cache_valid = true;
        }
        if (t->capture_bounce == b && t->d_cap && b + 1 < t->traceDepth) {       // K == 1 here (see ptx_render)
            // (K == 1, lane 0: segment 0 of the buffers)
            int32_t *gs = t->d_cap + 3 * (size_t)t->cap + nb, *ga = gs + chunk_cap + 1;
            hipLaunchKernelGGL(k_capture_prefix, dim3(1), dim3(64), 0, stream, bp.chunk, (int)chunk_cap, nb * gx_b, gs, ga);
            hipLaunchKernelGGL(k_capture, dim3(std::min(1024, (t->cap + 255) / 256)), dim3(256), 0, stream, bp.stage, bp.chunk, (int)chunk_cap,
                               nb * gx_b, gs, ga, t->cap, t->d_cap, t->d_cap_f, bp.idx16);
            HIPCHECK(hipMemcpyAsync(t->d_cap + 3 * (size_t)t->cap, totals(b, 1), sizeof(int32_t) * nb, hipMemcpyDeviceToDevice, stream));
            t->cap_filled = true;
        }
    }
    if (defer) {                                     // render-ahead: gather and statistics follow per iteration (finish_segment)
        t->ahead[lane].use_cache = use_cache;
        t->ahead[lane].dir_bins = batch_dir_bins; t->ahead[lane].ntab_bins = batch_ntab_bins;
        HIPCHECK(hipGetLastError());
        return PTX_OK;
    }
    if (prev_lane >= 0 && prev_lane != lane) HIPCHECK(hipStreamWaitEvent(stream, t->ev_chain[prev_lane], 0));      // the previous batch's gather + stats
    if (batched)
        hipLaunchKernelGGL(k_gather, dim3(std::min(2048, (t->tm.owned + 255) / 256)), dim3(256), 0, stream, t->tm, t->cam.resx, K,
                           t->seg_part, t->d_part + seg0 * t->seg_part, t->d_image, (uint32_t)t->cap);
    hipLaunchKernelGGL(k_stats, dim3(1), dim3(64), 0, stream, t->d_totals + seg0 * seg_totals, nb, t->traceDepth, 2 * nb, use_cache ? 1 : 0, K,
                       seg_totals, t->d_stats, t->d_stats + 64, batch_dir_bins, batch_ntab_bins, t->d_stats + 66);
    if (t->lanes > 1) HIPCHECK(hipEventRecord(t->ev_chain[lane], stream));
    HIPCHECK(hipGetLastError());
    t->iterations += K;
    (void)nsuper;
    return PTX_OK;
}

// ---- render-ahead (ptx_iterate) ---------------------------------------------------------------------------------------
// adds the time of the lane's batch, in proportion to the iterations that were taken from it, to the running total
void ahead_fold_time(ptx_tracer *t, int lane) {
    ptx_tracer::Ahead &a = t->ahead[lane];
    if (!a.timed || !a.unfolded || !a.count) { a.unfolded = 0; return; }
    float ms = 0.f;
    if (hipEventSynchronize(t->ev_ahead1[lane]) == hipSuccess && hipEventElapsedTime(&ms, t->ev_ahead0[lane], t->ev_ahead1[lane]) == hipSuccess)
        t->loop_ms_total += (double)ms * a.unfolded / a.count;
    a.unfolded = 0;
}

// forgets what was traced ahead (camera changed, another kind of call came in, the sequence jumped)
void ahead_discard(ptx_tracer *t) {
    for (int l = 1; l < MAX_LANES; l++) {
        if (t->ahead[l].valid || t->ahead[l].unfolded) ahead_fold_time(t, l);
        t->ahead[l].valid = false;
    }
    t->ahead_cur = t->ahead_nxt = -1;
    t->last_ahead_lane = -1;
}

bool ahead_possible(const ptx_tracer *t, int iter) {
    if (!t->render_ahead || t->lanes < 3 || t->kmax < 2 || t->ktiming || t->capture_bounce >= 0) return false;
    if (t->cache_active() && (!t->cache_valid || iter == 1)) return false;       // that call fills the first-bounce cache
    return true;
}

// traces iterations first .. first + kmax - 1 on `lane`, gathers nothing yet
int ahead_start(ptx_tracer *t, int lane, int first) {
    hipStream_t ls = t->lane_stream[lane];
    ahead_fold_time(t, lane);                            // the events are about to be re-recorded
    HIPCHECK(hipEventRecord(t->ev_fork, t->stream));     // after everything on the main stream so far: the cache fill, and the
    HIPCHECK(hipStreamWaitEvent(ls, t->ev_fork, 0));     // gathers that still read this lane's buffers
    HIPCHECK(hipEventRecord(t->ev_ahead0[lane], ls));
    int rc = enqueue_batch(t, first, t->kmax, 1, lane, -1, true);
    if (rc != PTX_OK) return rc;
    HIPCHECK(hipEventRecord(t->ev_ahead1[lane], ls));
    ptx_tracer::Ahead &a = t->ahead[lane];
    a.first = first; a.count = t->kmax; a.next = first; a.unfolded = 0; a.valid = true; a.timed = true;
    return PTX_OK;
}

// one iteration of a traced-ahead batch into the image: what the tail of enqueue_batch does for a whole batch
int ahead_finish_segment(ptx_tracer *t, int lane, int seg) {
    const ptx_tracer::Ahead &a = t->ahead[lane];
    const size_t sg = (size_t)lane * t->kmax + seg, seg_part = t->seg_part;
    HIPCHECK(hipStreamWaitEvent(t->stream, t->ev_ahead1[lane], 0));
    hipLaunchKernelGGL(k_gather, dim3(std::min(2048, (t->tm.owned + 255) / 256)), dim3(256), 0, t->stream, t->tm, t->cam.resx, 1,
                       seg_part, t->d_part + sg * seg_part, t->d_image, (uint32_t)t->cap);
    hipLaunchKernelGGL(k_stats, dim3(1), dim3(64), 0, t->stream, t->d_totals + sg * t->seg_totals, t->nbins, t->traceDepth, 2 * t->nbins,
                       a.use_cache ? 1 : 0, 1, t->seg_totals, t->d_stats, t->d_stats + 64, a.dir_bins, a.ntab_bins, t->d_stats + 66);
    HIPCHECK(hipGetLastError());
    t->iterations += 1;
    return PTX_OK;
}

// device scratch of the per-stage entry points: freed on every way out (a HIPCHECK that fails returns from the middle)
template <class T> struct DevBuf {
    T *p = nullptr;
    DevBuf() = default;
    DevBuf(const DevBuf &) = delete;
    DevBuf &operator=(const DevBuf &) = delete;
    ~DevBuf() { if (p) (void)hipFree(p); }
    operator T *() const { return p; }
};

}  // namespace

extern "C" {

const char *ptx_last_error(void) { return g_last_error.c_str(); }
void ptx_internal_set_error(const char *msg) { g_last_error = msg ? msg : ""; }

int ptx_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

void ptx_default_options(ptx_options *o) {
    memset(o, 0, sizeof *o);
    o->depth_of_field = 0; o->cache_first_bounce = 1; o->sort_by_material = 1; o->antialiasing = 1; o->bounding_box = 0;
    o->tile_rows = 0; o->tile_rank = 0; o->tile_world = 1; o->device = -1; o->batch = 0; o->no_lds_triangles = 0; o->apps_variant = 0; o->no_cull = 0;
}

// (kmax_cap > 0: at most that many iterations per launch set, whatever the rule or the option says; *oom = an allocation failed for
// want of memory and *kmax_used iterations per set were being allocated for)
static int create_tracer(int ngeoms, const ptx_geom *geoms, int nmaterials, const ptx_material *materials,
                         const ptx_camera *camera, int trace_depth, const ptx_options *options, float *external_image,
                         void *stream, ptx_tracer **out, int kmax_cap, bool *oom, int *kmax_used);

int ptx_create(int ngeoms, const ptx_geom *geoms, int nmaterials, const ptx_material *materials,
               const ptx_camera *camera, int trace_depth, const ptx_options *options, float *external_image,
               void *stream, ptx_tracer **out) {
    // an out-of-memory failure is retried with half the iterations per launch set (they only set how much is in flight, never what
    // is computed) until one iteration per set does not fit either
    int cap = 0;
    for (;;) {
        bool oom = false;
        int used = 0;
        const int rc = create_tracer(ngeoms, geoms, nmaterials, materials, camera, trace_depth, options, external_image, stream, out, cap, &oom, &used);
        if (rc == PTX_OK || !oom || used <= 1) return rc;
        (void)hipGetLastError();
        cap = used / 2;
    }
}

static int create_tracer(int ngeoms, const ptx_geom *geoms, int nmaterials, const ptx_material *materials,
                         const ptx_camera *camera, int trace_depth, const ptx_options *options, float *external_image,
                         void *stream, ptx_tracer **out, int kmax_cap, bool *oom, int *kmax_used) {
    if (!out) return set_error(PTX_ERR_INVALID, "out is NULL");
    *out = nullptr;
    if (ngeoms < 0 || nmaterials < 0 || (ngeoms && !geoms) || (nmaterials && !materials) || !camera)
        return set_error(PTX_ERR_INVALID, "missing scene arrays");
    if (camera->resolution[0] <= 0 || camera->resolution[1] <= 0) return set_error(PTX_ERR_INVALID, "resolution must be positive");
    if (trace_depth < 1) return set_error(PTX_ERR_UNSUPPORTED, "trace depth must be >= 1");
    if (nmaterials >= (1 << BIN_BITS) || ngeoms > 32767) return set_error(PTX_ERR_UNSUPPORTED, "more than 65535 materials or 32767 geoms");
    ptx_options opt;
    if (options) opt = *options; else ptx_default_options(&opt);
    if (opt.bounding_box) return set_error(PTX_ERR_UNSUPPORTED, "BOUNDING_BOX culling is off in the reference and not implemented");
    if (opt.tile_world < 1) opt.tile_world = 1;
    if (opt.tile_world > 1 && (opt.tile_rows < 1 || opt.tile_rank < 0 || opt.tile_rank >= opt.tile_world))
        return set_error(PTX_ERR_INVALID, "bad tile split");
    for (int i = 0; i < ngeoms; i++) {
        if (geoms[i].materialid < 0 || geoms[i].materialid >= nmaterials)
            return set_error(PTX_ERR_INVALID, "geom " + std::to_string(i) + " refers to a material that does not exist");
        if (geoms[i].faceSize < 0 || (geoms[i].faceSize > 0 && !geoms[i].faces)) return set_error(PTX_ERR_INVALID, "bad face array");
    }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0)
        return set_error(PTX_ERR_NODEVICE, "no HIP device available; this library has no CPU path");
    int dev = opt.device;
    if (dev < 0) HIPCHECK(hipGetDevice(&dev));
    if (dev >= ndev) return set_error(PTX_ERR_INVALID, "device ordinal out of range");
    HIPCHECK(hipSetDevice(dev));

    const KernelSet *ks = &g_kernels_here;
    if (opt.arith != PTX_ARITH_EXACT) {
        const void *tab = opt.arith == 1 ? (ptx_arith_kernels_1 ? ptx_arith_kernels_1() : nullptr)
                        : opt.arith == 2 ? (ptx_arith_kernels_2 ? ptx_arith_kernels_2() : nullptr) : nullptr;
        if (opt.arith < 0 || opt.arith > 2) return set_error(PTX_ERR_INVALID, "ptx_options.arith: 0 (exact), 1 (contracted) or 2 (fast)");
        if (!tab) return set_error(PTX_ERR_UNSUPPORTED, "this library was built without the code object of arithmetic level " + std::to_string(opt.arith));
        ks = static_cast<const KernelSet *>(tab);
        if (ks->arith != opt.arith) return set_error(PTX_ERR_HIP, "arithmetic code object mismatch");
    }
    ptx_tracer *t = new ptx_tracer;
    t->ks = ks;
    t->device = dev; t->opt = opt; t->traceDepth = trace_depth; t->ngeoms = ngeoms; t->nmats = nmaterials;
    camera_to_device(*camera, t->cam);
    const int W = t->cam.resx, H = t->cam.resy;
    // tile split: rows owned by this device
    t->tm.W = W; t->tm.H = H; t->tm.tile_world = opt.tile_world; t->tm.tile_rank = opt.tile_rank;
    t->tm.tile_rows = opt.tile_world > 1 ? opt.tile_rows : H;
    fastdiv_magic((uint32_t)W, t->tm.w_mul, t->tm.w_sh);
    fastdiv_magic((uint32_t)t->tm.tile_rows, t->tm.rows_mul, t->tm.rows_sh);
    int owned_rows = 0;
    if (opt.tile_world <= 1) owned_rows = H;
    else for (int y = 0; y < H; y++) if ((y / opt.tile_rows) % opt.tile_world == opt.tile_rank) owned_rows++;
    t->tm.owned = owned_rows * W;
    t->maxTiles = (std::max(t->tm.owned, 1) + TILE - 1) / TILE;
    t->cap = t->maxTiles * TILE;                          // whole tiles: the stage is written tile by tile
    t->nbins = opt.sort_by_material ? (nmaterials > 0 ? nmaterials : 1) : 1;
    t->maxBounces = trace_depth;
    hipDeviceProp_t prop;
    auto fail = [&](int code) { free_tracer(t); return code; };
#define HC(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) { if (e_ == hipErrorOutOfMemory) *oom = true; set_error(PTX_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e_)); return fail(PTX_ERR_HIP); } } while (0)
    HC(hipGetDeviceProperties(&prop, dev));
    {
        int per_cu = 16;                      // upper bound (sizes the per-workgroup tables); enqueue_batch picks 7, 8 or 16 per CU
        if (const char *e = getenv("PTX_DEBUG_WG_PER_CU")) { per_cu = std::max(1, atoi(e)); t->grid_forced = true; }      // tuning experiments only
        t->cus = prop.multiProcessorCount;
        // (also for K segments in one launch: a 1/8 tile's ten iterations as 10 x 101 workgroups of 10 tiles run 6 % FASTER than as
        // 10 x 128 of 8 -- a grid that does not quite fill the chip leaves room for the other launch set's kernel to start)
        t->grid = std::min(t->maxTiles, prop.multiProcessorCount * per_cu);
    }
    if (t->grid < 1) t->grid = 1;
    t->grid_seg = std::max(1, std::min(t->grid, t->maxTiles));      // what one segment (iteration) can use: sizes its tables
    if (stream) { t->stream = (hipStream_t)stream; t->own_stream = false; }
    else {
        // the main stream carries what a caller waits for (per-call gather, preview, frame read-back) while the other lanes
        // trace ahead in the background: it gets the highest priority, so that those short kernels are not queued behind
        // the bounce kernels' workgroups
        int prio_lo = 0, prio_hi = 0;
        if (hipDeviceGetStreamPriorityRange(&prio_lo, &prio_hi) != hipSuccess) { prio_lo = prio_hi = 0; (void)hipGetLastError(); }
        if (getenv("PTX_DEBUG_NO_PRIORITY")) prio_hi = prio_lo;
        HC(hipStreamCreateWithPriority(&t->stream, hipStreamNonBlocking, prio_hi));
        t->own_stream = true;
    }
    HC(hipEventCreate(&t->ev_start)); HC(hipEventCreate(&t->ev_stop));

    // scene upload (pathtraceInit, src/pathtrace.cu:111-146) -- flattened, no host struct is mutated
    std::vector<DGeom> hg((size_t)std::max(ngeoms, 1));
    std::vector<float> hfaces;
    std::vector<uint8_t> htex;
    for (int i = 0; i < ngeoms; i++) {
        const ptx_geom &g = geoms[i];
        DGeom &d = hg[i];
        memset(&d, 0, sizeof d);
        memcpy(d.xf, g.transform, 64); memcpy(d.inv, g.inverseTransform, 64); memcpy(d.invT, g.invTranspose, 64);
        d.type = g.type; d.materialid = g.materialid;
        d.faceStart = (int32_t)(hfaces.size() / 15); d.faceCount = g.faceSize;
        if (g.faceSize) hfaces.insert(hfaces.end(), g.faces, g.faces + (size_t)g.faceSize * 15);
        const ptx_texture *tx[4] = {&g.kd, &g.ks, &g.ke, &g.bump};
        for (int k = 0; k < 4; k++) {
            DTex &dt = d.tex[k];
            if (tx[k]->channels > 0 && tx[k]->image && tx[k]->width > 0 && tx[k]->height > 0) {
                if (tx[k]->channels < 3) { set_error(PTX_ERR_UNSUPPORTED, "textures need >= 3 channels"); return fail(PTX_ERR_UNSUPPORTED); }
                dt.w = tx[k]->width; dt.h = tx[k]->height; dt.ch = tx[k]->channels; dt.off = htex.size();
                t->uses_uv = 1;
                size_t nbytes = (size_t)dt.w * dt.h * dt.ch;
                htex.insert(htex.end(), tx[k]->image, tx[k]->image + nbytes);
            }
        }
    }
    // upload-time triangle table for the intersection loop: v0, e1 = v1 - v0, e2 = v2 - v0
    t->ntri = (int)(hfaces.size() / 15);
    std::vector<float> htri9((size_t)std::max(t->ntri, 1) * 9, 0.f);
    for (int j = 0; j < t->ntri; j++) {
        const float *f = &hfaces[(size_t)j * 15];
        float *o = &htri9[(size_t)j * 9];
        for (int k = 0; k < 3; k++) { o[k] = f[k]; o[3 + k] = f[5 + k] - f[k]; o[6 + k] = f[10 + k] - f[k]; }
    }
    // a BVH for every mesh with enough faces to repay it (pt_bvh.h); its nodes and leaf triangles stay in global memory
    {
        BvhBuild bb;
        std::vector<int32_t> roots((size_t)std::max(ngeoms, 1), -1), depths((size_t)std::max(ngeoms, 1), 0);
        std::vector<int32_t> wroots((size_t)std::max(ngeoms, 1), -1), wneeds((size_t)std::max(ngeoms, 1), 0);
        const bool use_wide = getenv("PTX_DEBUG_NO_WIDE_BVH") == nullptr;          // (A/B timing, tests of both walks)
        for (int i = 0; i < ngeoms; i++)
            if (hg[i].type == G_OBJ && hg[i].faceCount >= BVH_MIN_FACES && !opt.no_bvh) {
                roots[i] = bvhBuild(hfaces.data(), htri9.data(), hg[i].faceStart, hg[i].faceCount, bb, &depths[i], &wroots[i], &wneeds[i]);
                if (!use_wide) wroots[i] = -1;
                t->bvh_meshes++;
            }
        t->bvh_nodes = (int)(bb.nodes.size() / 2);
        {   // stack entries per lane for k_mesh: deepest tree + 1 (a tree of depth d needs d + 1), at least 8, at most BVH_STACK
            int deepest = 0;
            // (a tree walks its four-wide nodes when their walk fits BVH_STACK entries, else the binary tree front to back when
            // that fits, else the skip links: the stack is as long as the longest walk that is taken)
            for (int i = 0; i < ngeoms; i++) {
                if (roots[i] < 0) continue;
                if (wroots[i] >= 0 && wneeds[i] <= BVH_STACK) deepest = std::max(deepest, wneeds[i] - 1);
                else if (depths[i] < BVH_STACK) deepest = std::max(deepest, depths[i]);
            }
            t->bvh_stack = std::min(BVH_STACK, std::max(8, deepest + 1));
        }
        if (!t->bvh_meshes && !getenv("PTX_DEBUG_NO_CHUNKS"))          // spread the loops of small meshes over lanes (tileIntersect)
            for (int i = 0; i < ngeoms; i++)
                if (hg[i].type == G_OBJ) t->mesh_chunks = std::max(t->mesh_chunks, (hg[i].faceCount + MESH_CHUNK - 1) / MESH_CHUNK);
        if (t->bvh_meshes) {
            HC(hipMalloc(&t->d_bvh_nodes, sizeof(BvhQuad) * bb.nodes.size()));
            HC(hipMemcpy(t->d_bvh_nodes, bb.nodes.data(), sizeof(BvhQuad) * bb.nodes.size(), hipMemcpyHostToDevice));
            HC(hipMalloc(&t->d_bvh_tris, sizeof(float) * bb.tris.size()));
            HC(hipMemcpy(t->d_bvh_tris, bb.tris.data(), sizeof(float) * bb.tris.size(), hipMemcpyHostToDevice));
            HC(hipMalloc(&t->d_bvh_root, sizeof(int32_t) * roots.size()));
            HC(hipMemcpy(t->d_bvh_root, roots.data(), sizeof(int32_t) * roots.size(), hipMemcpyHostToDevice));
            HC(hipMalloc(&t->d_bvh_depth, sizeof(int32_t) * depths.size()));
            HC(hipMemcpy(t->d_bvh_depth, depths.data(), sizeof(int32_t) * depths.size(), hipMemcpyHostToDevice));
            if (use_wide && !bb.wide.empty()) {
                HC(hipMalloc(&t->d_bvh_wide, sizeof(BvhWide4) * bb.wide.size()));
                HC(hipMemcpy(t->d_bvh_wide, bb.wide.data(), sizeof(BvhWide4) * bb.wide.size(), hipMemcpyHostToDevice));
                HC(hipMalloc(&t->d_bvh_wroot, sizeof(int32_t) * wroots.size()));
                HC(hipMemcpy(t->d_bvh_wroot, wroots.data(), sizeof(int32_t) * wroots.size(), hipMemcpyHostToDevice));
                HC(hipMalloc(&t->d_bvh_wneed, sizeof(int32_t) * wneeds.size()));
                HC(hipMemcpy(t->d_bvh_wneed, wneeds.data(), sizeof(int32_t) * wneeds.size(), hipMemcpyHostToDevice));
            }
        }
    }
    // materials and geom tables go to LDS; the triangle tables join them when that leaves room for at least 2 workgroups
    // per CU (160 KB LDS, ~19 KB of sort buffers) -- otherwise they are read from global memory (L2-resident)
    t->tri_lds = (((size_t)nmaterials * 11 + (size_t)ngeoms * 58) * 4 <= 56 * 1024 && !opt.no_lds_triangles) ? 1 : 0;
    t->ntri_lds = (t->tri_lds && ((size_t)t->ntri * 27 + (size_t)nmaterials * 11 + (size_t)ngeoms * 58) * 4 <= 56 * 1024) ? t->ntri : 0;
    {   // k_bounce's dynamic LDS grows with the scene (tables) and with the number of material bins (ranking histogram):
        // check it against the device limit here, where the caller can be told, not at the first launch.  Step down first
        // (triangle tables, then all tables, to global memory: the plain per-ray loop over the geoms takes over), refuse
        // only what cannot run at all.
        const size_t limit = prop.sharedMemPerBlock;
        auto need = [&]() { return sizeof(int32_t) * (bounceLdsWords(t->tri_lds ? sceneTableWords(t->ntri_lds, nmaterials, ngeoms) : 0, t->nbins) + QUEUE_WORDS); };
        if (need() > limit && t->ntri_lds) t->ntri_lds = 0;
        if (need() > limit && t->tri_lds) t->tri_lds = 0;
        if (need() > limit) {
            set_error(PTX_ERR_UNSUPPORTED, "material sort over " + std::to_string(t->nbins) + " materials needs " + std::to_string(need()) +
                      " bytes of LDS per workgroup, the device offers " + std::to_string(limit) + ": render with sort_by_material = 0 (same image "
                      "only if the reference is built with SORT_BY_MATERIAL 0 too)");
            return fail(PTX_ERR_UNSUPPORTED);
        }
    }
    // per-geom table for the per-lane gathers (rows 0-2 of the three matrices) and conservative world boxes
    std::vector<float> hgtab((size_t)std::max(ngeoms, 1) * 40, 0.f), haabb((size_t)std::max(ngeoms, 1) * 8, 0.f);
    for (int i = 0; i < ngeoms; i++) {
        const DGeom &d = hg[i];
        float *o = &hgtab[(size_t)i * 40];
        const float *mats3[3] = {d.inv, d.xf, d.invT};
        for (int m = 0; m < 3; m++)
            for (int r = 0; r < 3; r++)
                for (int c = 0; c < 4; c++) o[m * 12 + r * 4 + c] = mats3[m][c * 4 + r];
        int32_t ints[4] = {d.type, d.materialid, d.faceStart, d.faceCount};
        memcpy(o + 36, ints, sizeof ints);
        float box[6];
        make_world_aabb(d, hfaces, box);
        for (int k = 0; k < 3; k++) { haabb[(size_t)i * 8 + k] = box[k]; haabb[(size_t)i * 8 + 4 + k] = box[3 + k]; }
        if (i < 32) {
            if (d.type == G_OBJ) t->mesh_bits |= 1u << i;
            else if (d.type == G_CUBE) t->cube_bits |= 1u << i;
            else if (d.type == G_SPHERE) t->sphere_bits |= 1u << i;
        }
    }
    t->cull = (t->tri_lds && ngeoms >= 1 && ngeoms <= 32 && !opt.no_cull) ? 1 : 0;
    {   // normals that do not depend on the ray, computed once with the device's own functions (compiled for the host with
        // the same flags: no contraction, IEEE divide and square root), so the kernels read what they would have computed
        std::vector<float> hfn((size_t)std::max(t->ntri, 1) * 3, 0.f), hcn((size_t)std::max(ngeoms, 1) * 18, 0.f);
        for (int i = 0; i < ngeoms; i++) {
            const DGeom &d = hg[i];
            if (d.type == G_OBJ) {
                if (d.tex[3].ch && i < 32) t->bump_bits |= 1u << i;
                for (int j = 0; j < d.faceCount; j++) {          // meshIntersectionTest, src/intersections.h:237-243
                    const float *tri = &hfaces[((size_t)d.faceStart + j) * 15];
                    const vec3 e1 = sub(ld3(tri + 5), ld3(tri)), e2 = sub(ld3(tri + 10), ld3(tri));
                    const vec3 objN = normalize(cross(e1, e2));
                    const vec3 n = normalize(multiplyMV(d.invT, objN, 0.f));
                    float *o = &hfn[((size_t)d.faceStart + j) * 3];
                    o[0] = n.x; o[1] = n.y; o[2] = n.z;
                }
            } else if (d.type == G_CUBE) {                       // boxIntersectionTest, src/intersections.h:86
                const float *invT = &hgtab[(size_t)i * 40 + 24];
                for (int side = 0; side < 6; side++) {
                    const int axis = side >> 1;
                    const float sgn = (side & 1) ? 1.f : -1.f;
                    const vec3 e = V3(axis == 0 ? sgn : 0.f, axis == 1 ? sgn : 0.f, axis == 2 ? sgn : 0.f);
                    const vec3 n = normalize(mulRows(invT, e, 0.0f));
                    float *o = &hcn[(size_t)i * 18 + side * 3];
                    o[0] = n.x; o[1] = n.y; o[2] = n.z;
                }
            }
        }
        if (ngeoms > 32) t->bump_bits = 0xffffffffu;          // (no per-geom bit beyond 32 geoms: such scenes do not take the tile path)
        HC(hipMalloc(&t->d_fnorm, sizeof(float) * hfn.size()));
        HC(hipMemcpy(t->d_fnorm, hfn.data(), sizeof(float) * hfn.size(), hipMemcpyHostToDevice));
        HC(hipMalloc(&t->d_cnorm, sizeof(float) * hcn.size()));
        HC(hipMemcpy(t->d_cnorm, hcn.data(), sizeof(float) * hcn.size(), hipMemcpyHostToDevice));
    }
    if (hfaces.empty()) hfaces.resize(15, 0.f);
    if (htex.empty()) htex.resize(16, 0);
    std::vector<DMaterial> hm((size_t)std::max(nmaterials, 1));
    static_assert(sizeof(DMaterial) == sizeof(ptx_material), "material layout");
    if (nmaterials) memcpy(hm.data(), materials, sizeof(DMaterial) * (size_t)nmaterials);
    {   // which records carry what (record_masks); off: more than 64 bins, no material, or PTX_DEBUG_NO_DIR_SKIP
        unsigned long long need = ~0ull, cubes = 0ull;
        const bool off = t->nbins > 64 || nmaterials < 1 || getenv("PTX_DEBUG_NO_DIR_SKIP") != nullptr;
        if (!off) {
            std::vector<int> gt((size_t)ngeoms), gm((size_t)ngeoms);
            for (int i = 0; i < ngeoms; i++) { gt[i] = hg[i].type; gm[i] = hg[i].materialid; }
            record_masks(nmaterials, hm.data(), ngeoms, gt.data(), gm.data(), opt.sort_by_material != 0, need, cubes);
        }
        t->dir_bins = off ? ~0ull : need;
        // (the code rides in bits 28-30 of the pixel slot; the tabulated normals are what the tile path's decodeKey reads)
        t->ntab_bins = (off || !t->cull || t->tm.owned >= (1 << 28) || getenv("PTX_DEBUG_NO_NORMAL_CODES")) ? 0ull : cubes;
    }
    HC(hipMalloc(&t->d_geoms, sizeof(DGeom) * hg.size()));
    HC(hipMemcpy(t->d_geoms, hg.data(), sizeof(DGeom) * hg.size(), hipMemcpyHostToDevice));
    HC(hipMalloc(&t->d_mats, sizeof(DMaterial) * hm.size()));
    HC(hipMemcpy(t->d_mats, hm.data(), sizeof(DMaterial) * hm.size(), hipMemcpyHostToDevice));
    HC(hipMalloc(&t->d_faces, sizeof(float) * hfaces.size()));
    HC(hipMemcpy(t->d_faces, hfaces.data(), sizeof(float) * hfaces.size(), hipMemcpyHostToDevice));
    HC(hipMalloc(&t->d_tri9, sizeof(float) * htri9.size()));
    HC(hipMemcpy(t->d_tri9, htri9.data(), sizeof(float) * htri9.size(), hipMemcpyHostToDevice));
    HC(hipMalloc(&t->d_gtab, sizeof(float) * hgtab.size()));
    HC(hipMemcpy(t->d_gtab, hgtab.data(), sizeof(float) * hgtab.size(), hipMemcpyHostToDevice));
    {   // (the device reads the boxes as centre + half extent; the host keeps corners for the camera tile masks)
        std::vector<float> hch(haabb.size(), 0.f);
        for (int i = 0; i < ngeoms; i++) world_box_centre_half(&haabb[(size_t)i * 8], &hch[(size_t)i * 8]);
        HC(hipMalloc(&t->d_aabb, sizeof(float) * hch.size()));
        HC(hipMemcpy(t->d_aabb, hch.data(), sizeof(float) * hch.size(), hipMemcpyHostToDevice));
    }
    t->h_aabb = haabb;
    HC(hipMalloc(&t->d_texels, htex.size()));
    HC(hipMemcpy(t->d_texels, htex.data(), htex.size(), hipMemcpyHostToDevice));

    const size_t npix = (size_t)W * H;
    if (external_image) { t->d_image = external_image; t->own_image = false; }
    else {
        HC(hipMalloc(&t->d_image, sizeof(float) * 3 * npix));
        HC(hipMemset(t->d_image, 0, sizeof(float) * 3 * npix));
        t->own_image = true;
    }
    // iterations per launch set: explicit option, or the rule below (about 24 M paths per set, within a quarter of the device's
    // memory).  With the first-bounce cache the iterations of a batch all start from the one cached bounce-0 stream
    int kmax = opt.batch;
    if (kmax <= 0) {
        // about 24 M paths per launch set, at least 12 iterations: 12 of a 1080p frame, up to 32 of a small frame or of one rank's tile
        // (1/8 of 1080p: 0.058 -> 0.048 ms per iteration with 32 instead of 8), fewer only where the streams of all launch sets in
        // flight (two stages of 19 words + radiance + the split search's keys and queue = 176 B per path and iteration) would pass
        // 64 GB of the 288.  Round 4: 12 instead of 5 at 3840x2160 (the rule was 16 GB with a guessed 400 B per path): every kernel of the
        // split bounce gets 2.4x the work per launch -- C5 1.23 -> 1.17 ms per iteration with round 3's kernels, and what the refilling
        // k_mesh needs: 1.4 M parked rays per launch instead of 0.6 M for the chip's 330 k lanes.
        // Round 5: the 64 GB are a ceiling, not a constant -- a quarter of what the device (a CPX / NPS partition, a GPU shared by
        // several ranks) has free or in total, whichever is less; and an allocation that still fails is retried with half the
        // iterations per set (ptx_create below) before the caller is told.
        const long long owned = std::max(t->tm.owned, 1);
        long long want = ((24LL << 20) + owned / 2) / owned;
        want = std::min<long long>(32, std::max<long long>(12, want));
        const long long nl = opt.lanes >= 1 ? std::min(opt.lanes, MAX_LANES) : 3;
        long long budget = 64LL << 30;
        size_t mem_free = 0, mem_total = 0;
        if (hipMemGetInfo(&mem_free, &mem_total) == hipSuccess && mem_total > 0) budget = std::min<long long>(budget, (long long)(std::min(mem_free, mem_total) / 4));
        else (void)hipGetLastError();
        if (const char *e = getenv("PTX_DEBUG_MEM_BUDGET_MB")) budget = std::max(1LL, atoll(e)) << 20;      // tests only
        kmax = (int)std::min<long long>(want, std::max<long long>(1, budget / (176LL * nl * owned)));
    }
    if (kmax_cap > 0 && kmax > kmax_cap) kmax = kmax_cap;      // (the retry after an allocation failed)
    if (kmax > 64) kmax = 64;
    // the per-tile prefix tables grow with bins x tiles x iterations in flight: keep them under 4 GiB by putting fewer
    // iterations into a launch set, then fewer launch sets in flight
    {
        const int want_lanes = opt.lanes >= 1 ? std::min(opt.lanes, MAX_LANES) : 3;
        auto counts_bytes = [&](int k, int l) { return sizeof(int32_t) * (2 * (size_t)t->nbins + 1) * t->maxTiles * (size_t)k * l; };
        while (counts_bytes(kmax, want_lanes) > (4ULL << 30) && kmax > 1) kmax /= 2;
        if (counts_bytes(kmax, want_lanes) > (4ULL << 30)) opt.lanes = t->opt.lanes = 1;
        if (counts_bytes(kmax, 1) > (4ULL << 30) && opt.lanes == 1) {
            set_error(PTX_ERR_UNSUPPORTED, "material sort over " + std::to_string(t->nbins) + " materials on " + std::to_string(t->maxTiles) +
                      " tiles needs more than 4 GiB of prefix tables: render with sort_by_material = 0");
            return fail(PTX_ERR_UNSUPPORTED);
        }
    }
    t->kmax = kmax;
    *kmax_used = kmax;
    // three launch sets in flight (one per stream) unless told otherwise: kernels of different sets overlap and
    // kernel tails are filled (C4, iterations per set x sets: 8 x 1 0.41, 8 x 2 0.30, 12 x 3 0.276, 12 x 4 0.31 ms per
    // iteration); also with one iteration per launch set, i.e. frames so large that only one fits the memory rule above
    // (7680 x 4320: 6.2 -> 4.75 ms per iteration); needs the per-iteration radiance buffers
    t->lanes = opt.lanes >= 1 ? std::min(opt.lanes, MAX_LANES) : 3;
    t->no_fast = getenv("PTX_DEBUG_NO_FAST") != nullptr;
    t->force_fast = getenv("PTX_DEBUG_FORCE_FAST") != nullptr;
    if (const char *e = getenv("PTX_DEBUG_TOTAL_WG_PER_CU")) t->dbg_total_wg_per_cu = std::max(0, atoi(e));
    if (const char *e = getenv("PTX_DEBUG_NSETS")) t->dbg_nsets = std::max(0, atoi(e));
    if (const char *e = getenv("PTX_DEBUG_EXTRA_LDS")) t->dbg_extra_lds = std::max(0, std::min(atoi(e), 32768)) & ~15;
    if (const char *e = getenv("PTX_DEBUG_MESH_WG_PER_CU")) t->dbg_mesh_wg_per_cu = std::max(0, atoi(e));
    if (const char *e = getenv("PTX_DEBUG_SPLIT_MIN")) t->split_min_paths = std::max(1LL, atoll(e));      // tuning experiments only
    if (t->lanes > 1) {
        HC(hipEventCreateWithFlags(&t->ev_fork, hipEventDisableTiming));
        for (int l = 0; l < t->lanes; l++) {
            if (l) {
                // every lane at the device's highest priority, like the main stream: measured, not reasoned -- sets of equal
                // priority at that level dispatch 2 % faster than at the default level (C4 wall 0.216 -> 0.211 ms), any
                // mix of levels lies in between, the veneer's per-call loop does not care
                int prio_lo = 0, prio = 0;
                if (hipDeviceGetStreamPriorityRange(&prio_lo, &prio) != hipSuccess || getenv("PTX_DEBUG_NO_PRIORITY")) { prio = 0; (void)hipGetLastError(); }
                if (const char *e = getenv("PTX_DEBUG_LANE_PRIO")) {          // tuning experiments only: p1,p2,... (HIP priority values)
                    const char *q = e;
                    for (int k = 1; k <= l && q; k++) { prio = atoi(q); q = strchr(q, ','); if (q) q++; }
                }
                HC(hipStreamCreateWithPriority(&t->lane_stream[l], hipStreamNonBlocking, prio));
            }
            HC(hipEventCreateWithFlags(&t->ev_join[l], hipEventDisableTiming));
            HC(hipEventCreateWithFlags(&t->ev_chain[l], hipEventDisableTiming));
            if (l) { HC(hipEventCreate(&t->ev_ahead0[l])); HC(hipEventCreate(&t->ev_ahead1[l])); }
        }
    }
    const size_t nseg = (size_t)kmax * t->lanes;
    t->field_stride = nseg * t->cap;
    const int nsoa = t->cache_active() ? 3 : 2;
    for (int k = 0; k < nsoa; k++) {
        const size_t stride = k == 2 ? (size_t)t->cap : t->field_stride;
        HC(hipMalloc(&t->d_fbuf[k], sizeof(float) * SOA_FLOATS * stride));
        HC(hipMalloc(&t->d_ibuf[k], sizeof(int32_t) * SOA_INTS * stride));
        carve(t->soa[k], t->d_fbuf[k], t->d_ibuf[k], stride);
    }
    t->seg_part = 3 * (size_t)t->cap + (size_t)t->cap / 4;      // per-iteration radiance of the OWNED pixels (slot-indexed), whole tiles, then the
                                                                // iteration's "lit" flags, a byte per slot (cap is a multiple of 256)
    if (nseg > 1) HC(hipMalloc(&t->d_part, sizeof(float) * t->seg_part * nseg));
    {   // split mesh search: worth it when some mesh is big enough for a BVH; needs the candidate masks (cull, <= 32 geoms: a
        // parked ray carries one bit per mesh whose box it reaches) and a queue entry per ray in the worst case
        int nmesh = 0;
        for (int i = 0; i < ngeoms; i++) nmesh += hg[i].type == G_OBJ ? 1 : 0;
        t->split_mesh = t->bvh_meshes > 0 && t->cull && !opt.no_mesh_split;
        if (getenv("PTX_DEBUG_FORCE_SPLIT")) t->split_mesh = t->cull && nmesh >= 1;      // timing experiments only
        if (t->split_mesh) {
            t->seg_items = (size_t)t->cap;
            HC(hipMalloc(&t->d_keys, sizeof(unsigned long long) * (size_t)t->cap * nseg));
            HC(hipMalloc(&t->d_items, sizeof(uint32_t) * t->seg_items * nseg));
            HC(hipMalloc(&t->d_item_count, sizeof(int32_t) * 2 * nseg));      // [nseg] counts (pass 1), [nseg] cursors (k_mesh)
            if (!getenv("PTX_DEBUG_NO_FIRST_FUSION")) {
                HC(hipMalloc(&t->d_tile_done, sizeof(int32_t) * (size_t)t->maxTiles * nseg));
                HC(hipMemset(t->d_tile_done, 0, sizeof(int32_t) * (size_t)t->maxTiles * nseg));
            }
        }
    }
    if (t->tri_lds) {   // the scene tables as k_bounce stages them (split: without the triangle tables), in one array: DScene::ldsblob
        const int nl = t->split_mesh ? 0 : t->ntri_lds;
        const size_t n9 = (size_t)nl * 9, n15 = (size_t)nl * 15, nm = (size_t)nmaterials * 11, ng = (size_t)ngeoms * GTAB_WORDS, nf = (size_t)nl * 3, nc = (size_t)ngeoms * 18;
        HC(hipMalloc(&t->d_ldsblob, sizeof(float) * std::max<size_t>(n9 + n15 + nm + ng + nf + nc, 4)));
        float *o = t->d_ldsblob;
        if (n9) HC(hipMemcpy(o, t->d_tri9, sizeof(float) * n9, hipMemcpyDeviceToDevice));
        o += n9;
        if (n15) HC(hipMemcpy(o, t->d_faces, sizeof(float) * n15, hipMemcpyDeviceToDevice));
        o += n15;
        if (nm) HC(hipMemcpy(o, t->d_mats, sizeof(float) * nm, hipMemcpyDeviceToDevice));
        o += nm;
        if (ng) HC(hipMemcpy(o, t->d_gtab, sizeof(float) * ng, hipMemcpyDeviceToDevice));
        o += ng;
        if (nf) HC(hipMemcpy(o, t->d_fnorm, sizeof(float) * nf, hipMemcpyDeviceToDevice));
        o += nf;
        if (nc) HC(hipMemcpy(o, t->d_cnorm, sizeof(float) * nc, hipMemcpyDeviceToDevice));
    }
    if (opt.apps_variant) {
        HC(hipMalloc(&t->d_albedo, sizeof(float) * 3 * npix));
        HC(hipMemset(t->d_albedo, 0, sizeof(float) * 3 * npix));
    }
    HC(hipMalloc(&t->d_counts, sizeof(int32_t) * (2 * (size_t)t->nbins + 1) * t->maxTiles * nseg));
    t->nsuper = (t->grid_seg + 63) / 64;
    HC(hipMalloc(&t->d_chunk, sizeof(int32_t) * 2 * 3 * (size_t)t->nbins * t->grid_seg * nseg));
    HC(hipMemset(t->d_chunk, 0, sizeof(int32_t) * 2 * 3 * (size_t)t->nbins * t->grid_seg * nseg));
    if (t->cache_active()) {
        HC(hipMalloc(&t->d_cache_chunk, sizeof(int32_t) * 3 * (size_t)t->nbins * t->grid_seg));
        HC(hipMemset(t->d_cache_chunk, 0, sizeof(int32_t) * 3 * (size_t)t->nbins * t->grid_seg));
    }
    t->seg_totals = 2 * (size_t)t->nbins * t->maxBounces * (1 + (size_t)t->nsuper);
    t->totals_bytes = sizeof(int32_t) * t->seg_totals * nseg;
    HC(hipMalloc(&t->d_totals, t->totals_bytes));
    HC(hipMemset(t->d_totals, 0, t->totals_bytes));
    t->d_super = t->d_totals + 2 * (size_t)t->nbins * t->maxBounces;
    HC(hipMalloc(&t->d_cache_totals, sizeof(int32_t) * 2 * (size_t)t->nbins));
    HC(hipMalloc(&t->d_cache_super, sizeof(int32_t) * 2 * (size_t)t->nbins * t->nsuper));
    HC(hipMalloc(&t->d_emit_count, sizeof(int32_t)));
    HC(hipMemset(t->d_emit_count, 0, sizeof(int32_t)));
    HC(hipMalloc(&t->d_emit_pix, sizeof(int32_t) * (size_t)t->cap));
    HC(hipMalloc(&t->d_emit_rgb, sizeof(float) * 3 * (size_t)t->cap));
#if defined(PT_STAMPS) || defined(PT_WGCLOCK)
    HC(hipMalloc(&t->d_stamps, sizeof(unsigned long long) * (48 + 2 * 64 * 4096 * 5)));
    HC(hipMemset(t->d_stamps, 0, sizeof(unsigned long long) * (48 + 2 * 64 * 4096 * 5)));
#endif
    HC(hipMalloc(&t->d_tile_geoms, sizeof(uint32_t) * (size_t)std::max(t->maxTiles, 1)));
    if (update_tile_geoms(t) != PTX_OK) return fail(PTX_ERR_HIP);
    HC(hipMalloc(&t->d_stats, sizeof(int64_t) * 69));
    HC(hipMemset(t->d_stats, 0, sizeof(int64_t) * 69));
    t->fence_slots = (uint32_t)t->cap;
    if (const char *e = getenv("PTX_DEBUG_FENCE_SLOTS")) t->fence_slots = (uint32_t)std::min<long long>(t->cap, std::max<long long>(1, atoll(e)));      // tests only
#undef HC
    *out = t;
    return PTX_OK;
}

int ptx_create_from_scene(const ptx_scene *s, const ptx_options *options, float *external_image, void *stream, ptx_tracer **out) {
    if (!s) return set_error(PTX_ERR_INVALID, "null scene");
    return ptx_create(ptx_scene_num_geoms(s), ptx_scene_geoms(s), ptx_scene_num_materials(s), ptx_scene_materials(s),
                      ptx_scene_camera(const_cast<ptx_scene *>(s)), ptx_scene_trace_depth(s), options, external_image, stream, out);
}

void ptx_destroy(ptx_tracer *t) { free_tracer(t); }

int ptx_set_camera(ptx_tracer *t, const ptx_camera *camera, int trace_depth) {
    if (!t || !camera) return set_error(PTX_ERR_INVALID, "null argument");
    if (camera->resolution[0] != t->cam.resx || camera->resolution[1] != t->cam.resy)
        return set_error(PTX_ERR_INVALID, "resolution is fixed at create (buffer sizes, src/pathtrace.cu:104-109)");
    if (trace_depth < 1 || trace_depth > t->maxBounces) return set_error(PTX_ERR_INVALID, "trace depth exceeds the depth given at create");
    {   // the reference-shaped loop sets the camera before every iteration (src/pathtrace.cu:434-436): the same values
        // again change nothing, so neither the first-bounce cache nor what was traced ahead is thrown away
        DCamera same;
        memcpy(&same, &t->cam, sizeof same);
        camera_to_device(*camera, same);
        if (trace_depth == t->traceDepth && memcmp(&same, &t->cam, sizeof same) == 0) return PTX_OK;
    }
    HIPCHECK(hipSetDevice(t->device));
    HIPCHECK(hipStreamSynchronize(t->stream));
    ahead_discard(t);
    camera_to_device(*camera, t->cam);
    t->traceDepth = trace_depth;
    t->cache_valid = false;
    return update_tile_geoms(t);
}

int ptx_reset_image(ptx_tracer *t) {
    if (!t) return set_error(PTX_ERR_INVALID, "null tracer");
    HIPCHECK(hipSetDevice(t->device));
    HIPCHECK(hipMemsetAsync(t->d_image, 0, sizeof(float) * 3 * (size_t)t->cam.resx * t->cam.resy, t->stream));
    HIPCHECK(hipMemsetAsync(t->d_stats, 0, sizeof(int64_t) * 69, t->stream));
    t->iterations = 0; t->loop_ms_total = 0.0; t->cache_valid = false;
    return PTX_OK;
}

// Experiment hook (PTX_DEBUG_PREQUEUE_US): one lane that holds the main stream for that long, so that the host has queued the whole run
// before its first kernel starts -- what a replayed launch graph would look like from the device's side.  The loop timer starts behind it.
namespace { __global__ void k_hold(long long ticks) { const long long t0 = wall_clock64(); while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(32); } }

int ptx_render(ptx_tracer *t, int iter_first, int count) { return ptx_render_strided(t, iter_first, count, 1); }

int ptx_render_strided(ptx_tracer *t, int iter_first, int count, int stride) {
    if (!t) return set_error(PTX_ERR_INVALID, "null tracer");
    if (stride < 1) return set_error(PTX_ERR_INVALID, "ptx_render_strided: stride must be >= 1");
    if (count <= 0) return PTX_OK;
    HIPCHECK(hipSetDevice(t->device));
    ahead_discard(t);
    if (t->timing_valid) {          // fold the previous batch's time into the running total before reusing events
        float ms = 0.f;
        HIPCHECK(hipEventSynchronize(t->ev_stop));
        HIPCHECK(hipEventElapsedTime(&ms, t->ev_start, t->ev_stop));
        t->loop_ms_total += ms;
        t->timing_valid = false;
    }
    if (const char *e = getenv("PTX_DEBUG_PREQUEUE_US")) hipLaunchKernelGGL(k_hold, dim3(1), dim3(64), 0, t->stream, (long long)atoi(e) * 100);      // (100 MHz)
    HIPCHECK(hipEventRecord(t->ev_start, t->stream));
    // per-kernel timing and the debug capture look at one launch set at a time
    // A run shorter than lanes x kmax iterations is cut into equal launch sets, one per lane, as long as each keeps at
    // least split_min_paths primary rays (below that the launches no longer fill the chip and overlap buys nothing).
    int kb = t->kmax;
    if (t->lanes > 1 && count < t->lanes * t->kmax) {
        const long long owned = std::max(t->tm.owned, 1);
        const int kmin = (int)std::min<long long>(t->kmax, (t->split_min_paths + owned - 1) / owned);
        // ... and into TWO sets rather than three while two can hold the call: a one-shot of three sets starts its third
        // late (the host issues the sets one after the other) and makes all of them smaller -- 20 iterations as 10 + 10
        // instead of 7 + 7 + 6: full frame equal, 1/2 tile -2 %, 1/4 and 1/8 tile -7 % (round 3's sweep; tools/gpu_tile_grid_sweep.py is its successor)
        int nsets = count <= 2 * t->kmax ? std::min(2, t->lanes) : t->lanes;
        if (t->dbg_nsets > 0) nsets = std::min(t->dbg_nsets, t->lanes);
        kb = std::min(t->kmax, std::max(kmin, (count + nsets - 1) / nsets));
    }
    const int nl = (t->lanes > 1 && !t->ktiming && t->capture_bounce < 0 && count > kb) ? t->lanes : 1;
    auto fork = [&]() -> int {                           // the other lanes start after what is on the main stream so far
        HIPCHECK(hipEventRecord(t->ev_fork, t->stream));
        for (int l = 1; l < nl; l++) HIPCHECK(hipStreamWaitEvent(t->lane_stream[l], t->ev_fork, 0));
        return PTX_OK;
    };
    if (nl > 1) { int rc = fork(); if (rc != PTX_OK) return rc; }
    int batch = 0, prev_lane = -1;
    bool used[MAX_LANES] = {};
    int first_set = 0;
    if (const char *e = getenv("PTX_DEBUG_FIRST_SET")) first_set = atoi(e);      // tuning experiments only
    for (int k = 0; k < count; batch++) {
        int K = std::min(nl > 1 ? kb : t->kmax, count - k);
        if (nl > 1 && first_set > 0 && count <= 2 * t->kmax) K = std::min(batch == 0 ? std::min(first_set, t->kmax) : t->kmax, count - k);
        if (t->capture_bounce >= 0) K = 1;                        // the debug capture looks at one stream
        if (t->cache_active() && (!t->cache_valid || iter_first + k * stride == 1)) K = 1;
        const int lane = nl > 1 ? batch % nl : 0;
        const bool fills = t->cache_active() && (!t->cache_valid || iter_first + k * stride == 1);
        int rc = enqueue_batch(t, iter_first + k * stride, K, stride, lane, nl > 1 ? prev_lane : -1);
        if (rc != PTX_OK) return rc;
        if (fills && nl > 1 && lane == 0) {              // the other lanes must not read the cache before it is written
            rc = fork();
            if (rc != PTX_OK) return rc;
        }
        used[lane] = true;
        prev_lane = lane;
        k += K;
    }
    for (int l = 1; l < nl; l++)
        if (used[l]) {
            HIPCHECK(hipEventRecord(t->ev_join[l], t->lane_stream[l]));
            HIPCHECK(hipStreamWaitEvent(t->stream, t->ev_join[l], 0));
        }
    HIPCHECK(hipEventRecord(t->ev_stop, t->stream));
    t->timing_valid = true;
    return PTX_OK;
}

int ptx_set_render_ahead(ptx_tracer *t, int on) {
    if (!t) return set_error(PTX_ERR_INVALID, "null tracer");
    if (!on) ahead_discard(t);
    t->render_ahead = on != 0;
    return PTX_OK;
}

int ptx_iterate(ptx_tracer *t, int iter) {
    if (!t) return set_error(PTX_ERR_INVALID, "null tracer");
    if (!ahead_possible(t, iter)) return ptx_render(t, iter, 1);
    HIPCHECK(hipSetDevice(t->device));
    if (t->ahead_cur < 0 || !t->ahead[t->ahead_cur].valid || t->ahead[t->ahead_cur].next != iter) {
        // nothing traced ahead for this iteration (first call, or the sequence jumped): start at it
        ahead_discard(t);
        if (t->timing_valid) {      // the previous ptx_render's time, before "previous operation" becomes this call
            float ms = 0.f;
            HIPCHECK(hipEventSynchronize(t->ev_stop));
            HIPCHECK(hipEventElapsedTime(&ms, t->ev_start, t->ev_stop));
            t->loop_ms_total += ms;
            t->timing_valid = false;
        }
        int rc = ahead_start(t, 1, iter);
        if (rc != PTX_OK) return rc;
        t->ahead_cur = 1;
    }
    const int lane = t->ahead_cur;
    ptx_tracer::Ahead &a = t->ahead[lane];
    // keep one batch ahead of the one being consumed, on the other lane
    if (t->ahead_nxt < 0) {
        const int other = lane == 1 ? 2 : 1;
        int rc = ahead_start(t, other, a.first + a.count);
        if (rc != PTX_OK) return rc;
        t->ahead_nxt = other;
    }
    int rc = ahead_finish_segment(t, lane, iter - a.first);
    if (rc != PTX_OK) return rc;
    a.next++; a.unfolded++;
    t->last_ahead_lane = lane;
    if (a.next == a.first + a.count) {                   // used up: on to the batch traced meanwhile
        a.valid = false;
        t->ahead_cur = t->ahead_nxt;
        t->ahead_nxt = -1;
    }
    return PTX_OK;
}

int ptx_synchronize(ptx_tracer *t) {
    if (!t) return set_error(PTX_ERR_INVALID, "null tracer");
    HIPCHECK(hipSetDevice(t->device));
    HIPCHECK(hipStreamSynchronize(t->stream));
    return PTX_OK;
}

int ptx_read_image(ptx_tracer *t, float *host_rgb) {
    if (!t || !host_rgb) return set_error(PTX_ERR_INVALID, "null argument");
    HIPCHECK(hipSetDevice(t->device));
    HIPCHECK(hipMemcpyAsync(host_rgb, t->d_image, sizeof(float) * 3 * (size_t)t->cam.resx * t->cam.resy, hipMemcpyDeviceToHost, t->stream));
    HIPCHECK(hipStreamSynchronize(t->stream));
    return PTX_OK;
}

// the accumulation buffer back from a checkpoint (W*H*3 floats, the layout ptx_read_image returns)
int ptx_write_image(ptx_tracer *t, const float *host_rgb) {
    if (!t || !host_rgb) return set_error(PTX_ERR_INVALID, "null argument");
    HIPCHECK(hipSetDevice(t->device));
    HIPCHECK(hipMemcpyAsync(t->d_image, host_rgb, sizeof(float) * 3 * (size_t)t->cam.resx * t->cam.resy, hipMemcpyHostToDevice, t->stream));
    HIPCHECK(hipStreamSynchronize(t->stream));
    return PTX_OK;
}

int ptx_read_albedo(ptx_tracer *t, float *host_rgb) {
    if (!t || !host_rgb) return set_error(PTX_ERR_INVALID, "null argument");
    if (!t->d_albedo) return set_error(PTX_ERR_INVALID, "the albedo AOV exists only with options.apps_variant = 1");
    HIPCHECK(hipSetDevice(t->device));
    HIPCHECK(hipMemcpyAsync(host_rgb, t->d_albedo, sizeof(float) * 3 * (size_t)t->cam.resx * t->cam.resy, hipMemcpyDeviceToHost, t->stream));
    HIPCHECK(hipStreamSynchronize(t->stream));
    return PTX_OK;
}

// sendToGPU + sendDenosiedImageToPBO (apps/src/pathtrace.cu:96-116,673-685): a finished host frame -> 8-bit, no /iter
int ptx_write_denoised_pbo(ptx_tracer *t, const float *host_rgb, uint8_t *host_rgba) {
    if (!t || !host_rgb || !host_rgba) return set_error(PTX_ERR_INVALID, "null argument");
    HIPCHECK(hipSetDevice(t->device));
    const size_t n = (size_t)t->cam.resx * t->cam.resy;
    float *d_in = nullptr; uchar4 *d_out = nullptr;
    HIPCHECK(hipMalloc(&d_in, sizeof(float) * 3 * n));
    HIPCHECK(hipMalloc(&d_out, 4 * n));
    HIPCHECK(hipMemcpyAsync(d_in, host_rgb, sizeof(float) * 3 * n, hipMemcpyHostToDevice, t->stream));
    hipLaunchKernelGGL(k_pbo, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, t->stream, d_out, (int)n, 1, d_in);   // iter = 1: pix / 1
    hipError_t e = hipMemcpyAsync(host_rgba, d_out, 4 * n, hipMemcpyDeviceToHost, t->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(t->stream);
    hipFree(d_in); hipFree(d_out);
    if (e != hipSuccess) return set_error(PTX_ERR_HIP, hipGetErrorString(e));
    return PTX_OK;
}

// sendToGPU itself (apps/src/pathtrace.cu:673-685): host frame in, the preview written to a DEVICE pbo, as the reference's
// mapped GL buffer is
int ptx_write_denoised_pbo_device(ptx_tracer *t, const float *host_rgb, void *device_uchar4) {
    if (!t || !host_rgb) return set_error(PTX_ERR_INVALID, "null argument");
    if (!device_uchar4) return PTX_OK;                    // NULL pbo => skip, like ptx_write_pbo_device
    HIPCHECK(hipSetDevice(t->device));
    const size_t n = (size_t)t->cam.resx * t->cam.resy;
    if (!t->d_denoised) HIPCHECK(hipMalloc(&t->d_denoised, sizeof(float) * 3 * n));      // dev_denoised_output, kept like the reference's
    HIPCHECK(hipMemcpyAsync(t->d_denoised, host_rgb, sizeof(float) * 3 * n, hipMemcpyHostToDevice, t->stream));
    hipLaunchKernelGGL(k_pbo, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, t->stream, (uchar4 *)device_uchar4, (int)n, 1, t->d_denoised);
    HIPCHECK(hipGetLastError());
    HIPCHECK(hipStreamSynchronize(t->stream));            // host_rgb may be reused by the caller right away
    return PTX_OK;
}

float *ptx_device_image(ptx_tracer *t) { return t ? t->d_image : nullptr; }
void *ptx_stream(ptx_tracer *t) { return t ? (void *)t->stream : nullptr; }
int ptx_owned_pixels(const ptx_tracer *t) { return t ? t->tm.owned : 0; }

int ptx_write_pbo_device(ptx_tracer *t, int iter, void *device_uchar4) {
    if (!t) return set_error(PTX_ERR_INVALID, "null tracer");
    if (!device_uchar4) return PTX_OK;                    // NULL pbo => skip (the reference would fault)
    HIPCHECK(hipSetDevice(t->device));
    int n = t->cam.resx * t->cam.resy;
    hipLaunchKernelGGL(k_pbo, dim3((n + 255) / 256), dim3(256), 0, t->stream, (uchar4 *)device_uchar4, n, iter, t->d_image);
    HIPCHECK(hipGetLastError());
    return PTX_OK;
}

int ptx_write_pbo(ptx_tracer *t, int iter, uint8_t *host_rgba) {
    if (!t || !host_rgba) return set_error(PTX_ERR_INVALID, "null argument");
    HIPCHECK(hipSetDevice(t->device));
    size_t n = (size_t)t->cam.resx * t->cam.resy;
    // the staging buffer stays with the tracer: hipFree would wait for the whole device, i.e. for work traced ahead
    if (!t->d_pbo) HIPCHECK(hipMalloc(&t->d_pbo, n * 4));
    int rc = ptx_write_pbo_device(t, iter, t->d_pbo);
    if (rc == PTX_OK) {
        hipError_t e = hipMemcpyAsync(host_rgba, t->d_pbo, n * 4, hipMemcpyDeviceToHost, t->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(t->stream);
        if (e != hipSuccess) rc = set_error(PTX_ERR_HIP, hipGetErrorString(e));
    }
    return rc;
}

double ptx_last_loop_ms(ptx_tracer *t) {
    if (!t) return 0.0;
    if (t->last_ahead_lane >= 0) {                       // a call served from a batch traced ahead: its share of that batch
        const ptx_tracer::Ahead &a = t->ahead[t->last_ahead_lane];
        float ms = 0.f;
        hipSetDevice(t->device);
        if (!a.count || hipEventSynchronize(t->ev_ahead1[t->last_ahead_lane]) != hipSuccess) return 0.0;
        if (hipEventElapsedTime(&ms, t->ev_ahead0[t->last_ahead_lane], t->ev_ahead1[t->last_ahead_lane]) != hipSuccess) return 0.0;
        return (double)ms / a.count;
    }
    if (!t->timing_valid) return 0.0;
    hipSetDevice(t->device);
    float ms = 0.f;
    if (hipEventSynchronize(t->ev_stop) != hipSuccess) return 0.0;
    if (hipEventElapsedTime(&ms, t->ev_start, t->ev_stop) != hipSuccess) return 0.0;
    return (double)ms;
}

int ptx_get_stats(ptx_tracer *t, ptx_stats *out) {
    if (!t || !out) return set_error(PTX_ERR_INVALID, "null argument");
    HIPCHECK(hipSetDevice(t->device));
    HIPCHECK(hipStreamSynchronize(t->stream));
    int64_t h[69];
    HIPCHECK(hipMemcpy(h, t->d_stats, sizeof h, hipMemcpyDeviceToHost));
    memset(out, 0, sizeof *out);
    out->bounces = t->traceDepth;
    for (int b = 0; b < 64 && b < t->traceDepth; b++) out->rays_per_bounce[b] = h[b];
    out->rays_total = h[64];
    out->fenced = h[65];
    out->stored_paths = h[66]; out->stored_with_direction = h[67]; out->stored_with_normal_code = h[68];
    for (int l = 1; l < MAX_LANES; l++) ahead_fold_time(t, l);
    out->loop_ms_total = t->loop_ms_total + (t->last_ahead_lane >= 0 ? 0.0 : ptx_last_loop_ms(t));
    out->iterations = t->iterations;
    return PTX_OK;
}

int ptx_get_stats_sized(ptx_tracer *t, void *out, size_t out_bytes) {
    if (!t || !out) return set_error(PTX_ERR_INVALID, "null argument");
    ptx_stats s;
    const int rc = ptx_get_stats(t, &s);
    if (rc != PTX_OK) return rc;
    memset(out, 0, out_bytes);
    memcpy(out, &s, std::min(out_bytes, sizeof s));
    return PTX_OK;
}

int ptx_abi_version(void) { return PTX_ABI_VERSION; }
size_t ptx_sizeof_options(void) { return sizeof(ptx_options); }
size_t ptx_sizeof_stats(void) { return sizeof(ptx_stats); }

// ---- per-stage entry points -----------------------------------------------------------------------------------
#define KAT_PROLOGUE                                                        \
    if (!t) return set_error(PTX_ERR_INVALID, "null tracer");               \
    HIPCHECK(hipSetDevice(t->device));                                      \
    HIPCHECK(hipStreamSynchronize(t->stream));

int ptx_kat_geom_test(ptx_tracer *t, int geom, int n, const float *rays6, float *out10) {
    KAT_PROLOGUE
    if (geom < 0 || geom >= t->ngeoms) return set_error(PTX_ERR_INVALID, "geom index out of range");
    if (n <= 0) return PTX_OK;
    DevBuf<float> d_in; DevBuf<float> d_out;
    HIPCHECK(hipMalloc(&d_in.p, sizeof(float) * 6 * (size_t)n));
    HIPCHECK(hipMalloc(&d_out.p, sizeof(float) * 10 * (size_t)n));
    HIPCHECK(hipMemcpy(d_in, rays6, sizeof(float) * 6 * (size_t)n, hipMemcpyHostToDevice));
    { const DScene sc = t->scene(); t->ks->kat_geom(dim3((n + 255) / 256), t->stream, &sc, geom, n, d_in, d_out); }
    HIPCHECK(hipStreamSynchronize(t->stream));
    HIPCHECK(hipMemcpy(out10, d_out, sizeof(float) * 10 * (size_t)n, hipMemcpyDeviceToHost));
    return PTX_OK;
}

int ptx_kat_obj_tri_test(ptx_tracer *t, int geom, int n, const float *rays6, float *out8) {
    KAT_PROLOGUE
    if (geom < 0 || geom >= t->ngeoms) return set_error(PTX_ERR_INVALID, "geom index out of range");
    if (n <= 0) return PTX_OK;
    DevBuf<float> d_in; DevBuf<float> d_out;
    HIPCHECK(hipMalloc(&d_in.p, sizeof(float) * 6 * (size_t)n));
    HIPCHECK(hipMalloc(&d_out.p, sizeof(float) * 8 * (size_t)n));
    HIPCHECK(hipMemcpy(d_in, rays6, sizeof(float) * 6 * (size_t)n, hipMemcpyHostToDevice));
    { const DScene sc = t->scene(); t->ks->kat_obj_tri(dim3((n + 255) / 256), t->stream, &sc, geom, n, d_in.p, d_out.p); }
    HIPCHECK(hipStreamSynchronize(t->stream));
    HIPCHECK(hipMemcpy(out8, d_out, sizeof(float) * 8 * (size_t)n, hipMemcpyDeviceToHost));
    return PTX_OK;
}

int ptx_kat_jittered_hemisphere(ptx_tracer *t, int n, const float *normals3, const int32_t *seeds3, int max_iter, float *out3) {
    KAT_PROLOGUE
    if (n <= 0) return PTX_OK;
    if (max_iter < 1) return set_error(PTX_ERR_INVALID, "max_iter must be >= 1");
    DevBuf<float> d_n, d_o; DevBuf<int32_t> d_s;
    HIPCHECK(hipMalloc(&d_n.p, 12 * (size_t)n)); HIPCHECK(hipMalloc(&d_o.p, 12 * (size_t)n)); HIPCHECK(hipMalloc(&d_s.p, 12 * (size_t)n));
    HIPCHECK(hipMemcpy(d_n, normals3, 12 * (size_t)n, hipMemcpyHostToDevice));
    HIPCHECK(hipMemcpy(d_s, seeds3, 12 * (size_t)n, hipMemcpyHostToDevice));
    t->ks->kat_jittered(dim3((n + 255) / 256), t->stream, n, d_n.p, d_s.p, max_iter, d_o.p);
    HIPCHECK(hipStreamSynchronize(t->stream));
    HIPCHECK(hipMemcpy(out3, d_o, 12 * (size_t)n, hipMemcpyDeviceToHost));
    return PTX_OK;
}

int ptx_kat_compute_intersections(ptx_tracer *t, int n, const void *paths44, void *isects32) {
    KAT_PROLOGUE
    if (n <= 0) return PTX_OK;
    DevBuf<HostPath> d_p; DevBuf<HostIsect> d_i;
    HIPCHECK(hipMalloc(&d_p.p, sizeof(HostPath) * (size_t)n));
    HIPCHECK(hipMalloc(&d_i.p, sizeof(HostIsect) * (size_t)n));
    HIPCHECK(hipMemcpy(d_p, paths44, sizeof(HostPath) * (size_t)n, hipMemcpyHostToDevice));
    { const DScene sc = t->scene(); t->ks->kat_intersect(dim3((n + 255) / 256), t->stream, &sc, n, d_p.p, d_i.p); }
    HIPCHECK(hipStreamSynchronize(t->stream));
    HIPCHECK(hipMemcpy(isects32, d_i, sizeof(HostIsect) * (size_t)n, hipMemcpyDeviceToHost));
    return PTX_OK;
}

int ptx_kat_tile_intersect(ptx_tracer *t, int n, const void *paths44, void *isects32, int split) {
    KAT_PROLOGUE
    if (n <= 0) return PTX_OK;
    if (!t->cull || !t->tri_lds) return set_error(PTX_ERR_UNSUPPORTED, "this scene does not take the tile path (candidate masks / LDS tables are off)");
    if (split && !t->d_bvh_root) return set_error(PTX_ERR_UNSUPPORTED, "no mesh of this scene has a BVH: nothing for the split mesh search to do");
    DevBuf<HostPath> d_p; DevBuf<HostIsect> d_i;
    HIPCHECK(hipMalloc(&d_p.p, sizeof(HostPath) * (size_t)n));
    HIPCHECK(hipMalloc(&d_i.p, sizeof(HostIsect) * (size_t)n));
    HIPCHECK(hipMemcpy(d_p, paths44, sizeof(HostPath) * (size_t)n, hipMemcpyHostToDevice));
    // the scene as enqueue_batch hands it to k_bounce (tables staged; split: without the triangle tables) and as k_mesh gets it
    DScene sc = t->scene();
    sc.tri_lds = t->tri_lds; sc.ntri_lds = (split || t->split_mesh) ? 0 : t->ntri_lds; sc.cull = t->cull;
    sc.ldsblob = (split || t->split_mesh) == t->split_mesh ? t->d_ldsblob : nullptr;      // (the blob is laid out for the tracer's own choice)
    DScene scg = t->scene();
    scg.bvh_stack = t->bvh_stack;
    const size_t lds = sizeof(int32_t) * (bounceLdsWords(sceneTableWords(sc.ntri_lds, t->nmats, t->ngeoms), 1) + (split ? (size_t)t->bvh_stack * TILE : 0));
    {   // ptx_create sized k_bounce's LDS against the device (stepping the tables down where needed); this kernel adds the walks'
        // stacks on top of the same layout, so it is checked here, where the caller can be told, not at the launch
        int lim = 0;
        HIPCHECK(hipDeviceGetAttribute(&lim, hipDeviceAttributeMaxSharedMemoryPerBlock, t->device));
        if (lds > (size_t)lim)
            return set_error(PTX_ERR_UNSUPPORTED, "ptx_kat_tile_intersect needs " + std::to_string(lds) + " bytes of LDS per workgroup (scene tables + " +
                             std::to_string(t->bvh_stack) + " stack entries per lane), the device offers " + std::to_string(lim));
    }
    const dim3 grid((unsigned)std::min(1024, (n + TILE - 1) / TILE));
    t->ks->kat_tile(split ? 1 : 0, grid, lds, t->stream, &sc, &scg, n, d_p.p, d_i.p, t->uses_uv);
    HIPCHECK(hipGetLastError());
    HIPCHECK(hipStreamSynchronize(t->stream));
    HIPCHECK(hipMemcpy(isects32, d_i, sizeof(HostIsect) * (size_t)n, hipMemcpyDeviceToHost));
    return PTX_OK;
}

int ptx_kat_shade(ptx_tracer *t, int iter, int n, const int32_t *idx, const void *isects32, void *paths44) {
    KAT_PROLOGUE
    if (n <= 0) return PTX_OK;
    DevBuf<HostPath> d_p; DevBuf<HostIsect> d_i; DevBuf<int32_t> d_x;
    HIPCHECK(hipMalloc(&d_p.p, sizeof(HostPath) * (size_t)n));
    HIPCHECK(hipMalloc(&d_i.p, sizeof(HostIsect) * (size_t)n));
    HIPCHECK(hipMalloc(&d_x.p, sizeof(int32_t) * (size_t)n));
    HIPCHECK(hipMemcpy(d_p, paths44, sizeof(HostPath) * (size_t)n, hipMemcpyHostToDevice));
    HIPCHECK(hipMemcpy(d_i, isects32, sizeof(HostIsect) * (size_t)n, hipMemcpyHostToDevice));
    HIPCHECK(hipMemcpy(d_x, idx, sizeof(int32_t) * (size_t)n, hipMemcpyHostToDevice));
    { const DScene sc = t->scene(); t->ks->kat_shade(dim3((n + 255) / 256), t->stream, &sc, iter, n, d_x.p, d_i.p, d_p.p); }
    HIPCHECK(hipStreamSynchronize(t->stream));
    HIPCHECK(hipMemcpy(paths44, d_p, sizeof(HostPath) * (size_t)n, hipMemcpyDeviceToHost));
    return PTX_OK;
}

int ptx_kat_generate(ptx_tracer *t, int iter, void *paths44) {
    KAT_PROLOGUE
    int n = t->cam.resx * t->cam.resy;
    DevBuf<HostPath> d_p;
    HIPCHECK(hipMalloc(&d_p.p, sizeof(HostPath) * (size_t)n));
    t->ks->kat_generate(dim3((n + 255) / 256), t->stream, &t->cam, iter, t->traceDepth, t->opt.antialiasing, t->opt.depth_of_field, d_p.p);
    HIPCHECK(hipStreamSynchronize(t->stream));
    HIPCHECK(hipMemcpy(paths44, d_p, sizeof(HostPath) * (size_t)n, hipMemcpyDeviceToHost));
    return PTX_OK;
}

int ptx_kat_libm(ptx_tracer *t, int n, const float *x, float *sin_out, float *cos_out, const double *pw_in,
                 double *pow5_out, const float *powf_xy, float *powf_out) {
    KAT_PROLOGUE
    if (n <= 0) return PTX_OK;
    DevBuf<float> dx, ds, dc, dxy, dpo; DevBuf<double> dpw, dp5;
    HIPCHECK(hipMalloc(&dx.p, 4 * (size_t)n)); HIPCHECK(hipMalloc(&ds.p, 4 * (size_t)n)); HIPCHECK(hipMalloc(&dc.p, 4 * (size_t)n));
    HIPCHECK(hipMalloc(&dxy.p, 8 * (size_t)n)); HIPCHECK(hipMalloc(&dpo.p, 4 * (size_t)n));
    HIPCHECK(hipMalloc(&dpw.p, 8 * (size_t)n)); HIPCHECK(hipMalloc(&dp5.p, 8 * (size_t)n));
    HIPCHECK(hipMemcpy(dx, x, 4 * (size_t)n, hipMemcpyHostToDevice));
    HIPCHECK(hipMemcpy(dpw, pw_in, 8 * (size_t)n, hipMemcpyHostToDevice));
    HIPCHECK(hipMemcpy(dxy, powf_xy, 8 * (size_t)n, hipMemcpyHostToDevice));
    t->ks->kat_libm(dim3((n + 255) / 256), t->stream, n, dx.p, ds.p, dc.p, dpw.p, dp5.p, dxy.p, dpo.p);
    HIPCHECK(hipStreamSynchronize(t->stream));
    HIPCHECK(hipMemcpy(sin_out, ds, 4 * (size_t)n, hipMemcpyDeviceToHost));
    HIPCHECK(hipMemcpy(cos_out, dc, 4 * (size_t)n, hipMemcpyDeviceToHost));
    HIPCHECK(hipMemcpy(pow5_out, dp5, 8 * (size_t)n, hipMemcpyDeviceToHost));
    HIPCHECK(hipMemcpy(powf_out, dpo, 4 * (size_t)n, hipMemcpyDeviceToHost));
    return PTX_OK;
}

int ptx_kat_fast_exact(ptx_tracer *t, int64_t mismatches[3]) {
    KAT_PROLOGUE
    if (!mismatches) return set_error(PTX_ERR_INVALID, "null argument");
    DevBuf<unsigned long long> d;
    HIPCHECK(hipMalloc(&d.p, 3 * sizeof(unsigned long long)));
    HIPCHECK(hipMemsetAsync(d.p, 0, 3 * sizeof(unsigned long long), t->stream));
    hipLaunchKernelGGL(k_kat_fast_exact, dim3(t->cus * 8), dim3(256), 0, t->stream, d.p);
    HIPCHECK(hipGetLastError());
    HIPCHECK(hipStreamSynchronize(t->stream));
    unsigned long long h[3];
    HIPCHECK(hipMemcpy(h, d.p, sizeof h, hipMemcpyDeviceToHost));
    for (int k = 0; k < 3; k++) mismatches[k] = (int64_t)h[k];
    return PTX_OK;
}

int ptx_set_kernel_timing(ptx_tracer *t, int on) {
    if (!t) return set_error(PTX_ERR_INVALID, "null tracer");
    HIPCHECK(hipSetDevice(t->device));
    HIPCHECK(hipStreamSynchronize(t->stream));
    ahead_discard(t);
    t->ktiming = on != 0;
    t->kev_used = 0;
    return PTX_OK;
}

int ptx_get_kernel_times(ptx_tracer *t, double ms_by_kind[4], int64_t launches_by_kind[4]) {
    if (!t || !ms_by_kind || !launches_by_kind) return set_error(PTX_ERR_INVALID, "null argument");
    HIPCHECK(hipSetDevice(t->device));
    HIPCHECK(hipStreamSynchronize(t->stream));
    for (int k = 0; k < 4; k++) { ms_by_kind[k] = 0.0; launches_by_kind[k] = 0; }
    for (size_t i = 0; i + 1 < t->kev_used; i += 2) {
        float ms = 0.f;
        HIPCHECK(hipEventElapsedTime(&ms, t->kev[i], t->kev[i + 1]));
        int kind = t->kev_kind[i / 2];
        ms_by_kind[kind] += ms; launches_by_kind[kind]++;
    }
    t->kev_used = 0;
    return PTX_OK;
}

// diagnostic build (-DPT_STAMPS): cycles per phase of k_bounce summed over waves: [0..4] first bounce, [8..12] later bounces
// CPU-only check of the mesh BVH: builds the tree of `nfaces` faces and searches `nrays` object-space rays (origin,
// direction; the direction is normalised the way meshIntersectionTest does) with the tree and with the plain loop.
static int64_t g_bvh_visits[8] = {0, 0, 0, 0, 0, 0, 0, 0};
int ptx_debug_bvh_check(const float *faces15, int nfaces, const float *rays6, int nrays, int32_t *face_loop, float *t_loop,
                        int32_t *face_bvh, float *t_bvh, int64_t *stats4) {
    if (!faces15 || !rays6 || !face_loop || !t_loop || !face_bvh || !t_bvh || nfaces < 1 || nrays < 0)
        return set_error(PTX_ERR_INVALID, "ptx_debug_bvh_check: bad argument");
    std::vector<float> tri9((size_t)nfaces * 9);
    for (int j = 0; j < nfaces; j++) {
        const float *f = faces15 + (size_t)j * 15;
        float *o = &tri9[(size_t)j * 9];
        for (int k = 0; k < 3; k++) { o[k] = f[k]; o[3 + k] = f[5 + k] - f[k]; o[6 + k] = f[10 + k] - f[k]; }
    }
    BvhBuild bb;
    int depth = 0;
    int wroot = -1, wneed = 0;
    const int root = bvhBuild(faces15, tri9.data(), 0, nfaces, bb, &depth, &wroot, &wneed);
    std::vector<int32_t> wstack((size_t)std::max(wneed, 1) + 1, 0x7fffffff);      // (+ a guard word: the walk must never reach it)
    long long visited = 0, visited_ordered = 0, visited_wide = 0, mismatches = 0, group_max = 0, sum_group_max = 0, groups = 0, tris_wide = 0;
    for (int i = 0; i < nrays; i++) {
        const vec3 o = V3(rays6[i * 6 + 0], rays6[i * 6 + 1], rays6[i * 6 + 2]);
        const vec3 d = normalize(V3(rays6[i * 6 + 3], rays6[i * 6 + 4], rays6[i * 6 + 5]));
        int f0, f1, vis = 0;
        float b0, b1;
        t_loop[i] = loopNearestHost(faces15, tri9.data(), nfaces, o, d, f0);
        t_bvh[i] = bvhNearest(bb.nodes.data(), bb.tris.data(), root, o, d, f1, b0, b1, &vis);
        if (depth < BVH_STACK) {                // the front-to-back search must agree with the skip-link one
            int f2, vis2 = 0;
            float c0, c1;
            int32_t stack[BVH_STACK];
            const float t2 = bvhNearestOrdered(bb.nodes.data(), bb.tris.data(), root, o, d, f2, c0, c1, stack, 1, &vis2);
            visited_ordered += vis2;
            if (f2 != f1 || memcmp(&t2, &t_bvh[i], 4) != 0 || (f1 >= 0 && (memcmp(&c0, &b0, 4) != 0 || memcmp(&c1, &b1, 4) != 0))) mismatches++;
        }
        if (wroot >= 0) {                       // ... and so must the walk over the four-wide nodes
            int f3, vis3 = 0;
            float e0, e1;
            const float t3 = bvhNearestWide(bb.nodes.data(), bb.wide.data(), bb.tris.data(), root, wroot, o, d, f3, e0, e1, wstack.data(), 1, &vis3);
            visited_wide += vis3 & 0xffff;
            tris_wide += vis3 >> 16;
            group_max = std::max(group_max, (long long)(vis3 & 0xffff) / 4);
            if (i % 64 == 63 || i == nrays - 1) { sum_group_max += group_max; group_max = 0; groups++; }
            if (wstack[(size_t)std::max(wneed, 1)] != 0x7fffffff) mismatches += 1000000;      // the walk overran the stack bound the builder computed
            if (f3 != f1 || memcmp(&t3, &t_bvh[i], 4) != 0 || (f1 >= 0 && (memcmp(&e0, &b0, 4) != 0 || memcmp(&e1, &b1, 4) != 0))) mismatches++;
            {   // the same steps under the schedule of k_mesh's refilling waves: one node or ONE triangle per turn
                WideWalk w;
                wideStart(w, bb.nodes[2 * (size_t)root], bb.nodes[2 * (size_t)root + 1], wroot, o, d);
                while (w.n != WIDE_DONE) {
                    if (w.n >= 0) wideNodeStep(w, bb.wide.data(), wstack.data(), 1);
                    else wideLeafStep<true>(w, bb.tris.data(), wstack.data(), 1);
                }
                if (wstack[(size_t)std::max(wneed, 1)] != 0x7fffffff) mismatches += 1000000;
                if (w.face != f1 || memcmp(&w.tmin, &t_bvh[i], 4) != 0 || (f1 >= 0 && (memcmp(&w.b0, &b0, 4) != 0 || memcmp(&w.b1, &b1, 4) != 0))) mismatches++;
            }
        }
        face_loop[i] = f0; face_bvh[i] = f1;
        visited += vis;
    }
    if (stats4) { stats4[0] = (int64_t)(bb.nodes.size() / 2); stats4[1] = (int64_t)(bb.tris.size() / BVH_TRI); stats4[2] = visited; stats4[3] = mismatches; }
    g_bvh_visits[0] = visited; g_bvh_visits[1] = visited_ordered; g_bvh_visits[2] = visited_wide / 4; g_bvh_visits[3] = wneed;
    g_bvh_visits[4] = sum_group_max; g_bvh_visits[5] = groups; g_bvh_visits[6] = tris_wide;
    return PTX_OK;
}

// node visits of the last ptx_debug_bvh_check: skip-link walk, front-to-back binary walk, four-wide walk (nodes), wide stack need,
// CPU-only: ptx_create's rule for what a stored path's record carries (record_masks), for nmaterials <= 64 materials and ngeoms geoms
// given by type and material: masks[0] = dir_bins, masks[1] = ntab_bins (before the conditions of a particular tracer: candidate masks
// on, fewer than 2^28 owned pixels).
int ptx_debug_record_masks(int nmaterials, const ptx_material *materials, int ngeoms, const int32_t *geom_type, const int32_t *geom_material,
                           int sort_by_material, uint64_t masks[2]) {
    if (nmaterials < 1 || nmaterials > 64 || !materials || ngeoms < 0 || (ngeoms && (!geom_type || !geom_material)) || !masks)
    { set_error(PTX_ERR_INVALID, "ptx_debug_record_masks: bad argument"); return -1; }
    unsigned long long d = 0, n = 0;
    record_masks(nmaterials, reinterpret_cast<const DMaterial *>(materials), ngeoms, geom_type, geom_material, sort_by_material != 0, d, n);
    masks[0] = d; masks[1] = n;
    return 0;
}

// CPU-only: the candidate pre-test's table (world_box_centre_half) for n corner boxes (lo xyz, hi xyz): 8 floats each = centre xyz, 0,
// half extent xyz, 0 -- what cullMask reads on the device.
int ptx_debug_cull_boxes(int n, const float *boxes6, float *centre_half8) {
    if (n < 0 || (n && (!boxes6 || !centre_half8))) { set_error(PTX_ERR_INVALID, "ptx_debug_cull_boxes: bad argument"); return -1; }
    for (int g = 0; g < n; g++) {
        const float lohi[8] = {boxes6[g * 6], boxes6[g * 6 + 1], boxes6[g * 6 + 2], 0.f, boxes6[g * 6 + 3], boxes6[g * 6 + 4], boxes6[g * 6 + 5], 0.f};
        world_box_centre_half(lohi, centre_half8 + (size_t)g * 8);
    }
    return n;
}

// sum over groups of 64 consecutive rays of the longest four-wide walk in the group, number of groups, triangles the four-wide walk tested
// CPU-only: the per-tile geom masks of the camera-ray bounce (update_tile_geoms) for a camera, a tile split and a list of world boxes
// (6 floats each: lo xyz, hi xyz), without a tracer or a device.  masks_out[tile], tiles of 256 owned pixels; returns the number of tiles
// (negative: bad argument).
int ptx_debug_tile_geoms(const ptx_camera *camera, int ngeoms, const float *boxes6, int depth_of_field, int tile_rows, int tile_rank, int tile_world,
                         uint32_t *masks_out, int max_tiles) {
    if (!camera || !boxes6 || !masks_out || ngeoms < 1 || ngeoms > 32 || camera->resolution[0] < 1 || camera->resolution[1] < 1)
    { set_error(PTX_ERR_INVALID, "ptx_debug_tile_geoms: bad argument"); return -1; }
    DCamera cam;
    camera_to_device(*camera, cam);
    TileMap tm{};
    const int W = cam.resx, H = cam.resy;
    tm.W = W; tm.H = H; tm.tile_world = tile_world < 1 ? 1 : tile_world; tm.tile_rank = tile_rank; tm.tile_rows = tm.tile_world > 1 ? tile_rows : H;
    if (tm.tile_world > 1 && (tile_rows < 1 || tile_rank < 0 || tile_rank >= tm.tile_world)) return -1;
    int owned_rows = 0;
    for (int y = 0; y < H; y++) if (tm.tile_world <= 1 || (y / tm.tile_rows) % tm.tile_world == tm.tile_rank) owned_rows++;
    tm.owned = owned_rows * W;
    const int ntiles = (std::max(tm.owned, 1) + TILE - 1) / TILE;
    if (ntiles > max_tiles) return -1;
    std::vector<float> a8((size_t)ngeoms * 8, 0.f);
    for (int g = 0; g < ngeoms; g++)
        for (int k = 0; k < 3; k++) { a8[(size_t)g * 8 + k] = boxes6[g * 6 + k]; a8[(size_t)g * 8 + 4 + k] = boxes6[g * 6 + 3 + k]; }
    std::vector<uint32_t> masks;
    tile_geom_masks(cam, tm, ntiles, ngeoms, a8.data(), depth_of_field != 0, masks);
    memcpy(masks_out, masks.data(), sizeof(uint32_t) * (size_t)ntiles);
    return ntiles;
}

int ptx_debug_bvh_visits(int64_t out8[8]) {
    if (!out8) return set_error(PTX_ERR_INVALID, "null argument");
    for (int k = 0; k < 8; k++) out8[k] = g_bvh_visits[k];
    return PTX_OK;
}

int ptx_debug_read_stamps(ptx_tracer *t, unsigned long long out48[48]) {
    if (!t || !out48) return set_error(PTX_ERR_INVALID, "null argument");
    memset(out48, 0, sizeof(unsigned long long) * 48);
    if (!t->d_stamps) return PTX_OK;
    HIPCHECK(hipSetDevice(t->device));
    HIPCHECK(hipStreamSynchronize(t->stream));
    HIPCHECK(hipMemcpy(out48, t->d_stamps, sizeof(unsigned long long) * 48, hipMemcpyDeviceToHost));
    {   // -DPT_WGCLOCK: the per-workgroup wall-clock slots summed per kind into [32..36] (first bounce) and [40..44] (later bounces)
        std::vector<unsigned long long> w((size_t)2 * 64 * 4096 * 5);
        HIPCHECK(hipMemcpy(w.data(), t->d_stamps + 48, sizeof(unsigned long long) * w.size(), hipMemcpyDeviceToHost));
        for (int kind = 0; kind < 2; kind++)
            for (size_t k = 0; k < (size_t)64 * 4096; k++)
                for (int f = 0; f < 5; f++) out48[32 + kind * 8 + f] += w[((size_t)kind * 64 * 4096 + k) * 5 + f];
    }
    HIPCHECK(hipMemset(t->d_stamps, 0, sizeof(unsigned long long) * (48 + 2 * 64 * 4096 * 5)));
    return PTX_OK;
}

// Debug: how many workgroups of the specialised later-bounce kernel the runtime says fit a CU with `lds_bytes` of dynamic LDS each
// (0 = what this tracer launches it with); negative = error.
int ptx_debug_bounce_occupancy(ptx_tracer *t, int lds_bytes) {
    if (!t) return -1;
    if (hipSetDevice(t->device) != hipSuccess) return -1;
    const int ntri_lds = t->split_mesh ? 0 : t->ntri_lds;
    const int triWords = t->tri_lds ? sceneTableWords(ntri_lds, t->nmats, t->ngeoms) : 0;
    size_t lds = lds_bytes > 0 ? (size_t)lds_bytes : sizeof(int32_t) * (bounceLdsWords(triWords, t->nbins) - (17 - REC_ROWS_FAST0) * TILE);
    int n = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, k_bounce<false, 0, true>, TILE, lds) != hipSuccess) return -2;
    return n;
}

int ptx_debug_set_capture(ptx_tracer *t, int bounce) {
    if (!t) return set_error(PTX_ERR_INVALID, "null tracer");
    HIPCHECK(hipSetDevice(t->device));
    t->capture_bounce = bounce;
    t->cap_filled = false;
    if (bounce >= 0 && !t->d_cap) {
        // pix, stream index, material|geom [cap each], the bounce's totals [nbins], run prefixes of the capture [2][nbins x grid_seg + 1]
        HIPCHECK(hipMalloc(&t->d_cap, sizeof(int32_t) * (3 * (size_t)t->cap + (size_t)t->nbins + 2 * ((size_t)t->nbins * t->grid_seg + 1))));
        HIPCHECK(hipMalloc(&t->d_cap_f, sizeof(float) * SOA_LOGICAL_FLOATS * (size_t)t->cap));
    }
    return PTX_OK;
}

int ptx_debug_read_stream(ptx_tracer *t, int *n_out, int32_t *pixel_index, int32_t *stream_idx, int32_t *material,
                          float *fields14, int cap) {
    if (!t || !n_out) return set_error(PTX_ERR_INVALID, "null argument");
    HIPCHECK(hipSetDevice(t->device));
    HIPCHECK(hipStreamSynchronize(t->stream));
    *n_out = 0;
    if (!t->cap_filled) return set_error(PTX_ERR_INVALID, "nothing captured");
    std::vector<int32_t> tot((size_t)t->nbins);
    HIPCHECK(hipMemcpy(tot.data(), t->d_cap + 3 * (size_t)t->cap, sizeof(int32_t) * tot.size(), hipMemcpyDeviceToHost));
    int n = 0;
    for (int v : tot) n += v;
    *n_out = n;
    int m = n < cap ? n : cap;
    if (m > 0) {
        HIPCHECK(hipMemcpy(pixel_index, t->d_cap, sizeof(int32_t) * (size_t)m, hipMemcpyDeviceToHost));
        if (t->tm.tile_world > 1)             // paths carry their slot among the owned pixels: report the pixel (x + y*W)
            for (int k = 0; k < m; k++) {
                const int slot = pixel_index[k] & 0x0fffffff, r = slot / t->tm.W, x = slot - r * t->tm.W, blk = r / t->tm.tile_rows;
                pixel_index[k] = x + ((blk * t->tm.tile_world + t->tm.tile_rank) * t->tm.tile_rows + (r - blk * t->tm.tile_rows)) * t->tm.W;
            }
        HIPCHECK(hipMemcpy(stream_idx, t->d_cap + t->cap, sizeof(int32_t) * (size_t)m, hipMemcpyDeviceToHost));
        std::vector<int32_t> mg((size_t)m);
        HIPCHECK(hipMemcpy(mg.data(), t->d_cap + 2 * (size_t)t->cap, sizeof(int32_t) * (size_t)m, hipMemcpyDeviceToHost));
        for (int k = 0; k < m; k++) material[k] = mg[k] & 0xffff;
        if (fields14)       // 14 rows of m floats: px py pz dx dy dz cr cg cb nx ny nz u v
            for (int f = 0; f < SOA_LOGICAL_FLOATS; f++)
                HIPCHECK(hipMemcpy(fields14 + (size_t)f * m, t->d_cap_f + (size_t)f * t->cap, sizeof(float) * (size_t)m, hipMemcpyDeviceToHost));
    }
    return PTX_OK;
}

}  // extern "C"

#endif  // PT_ARITH == 0: the host side
