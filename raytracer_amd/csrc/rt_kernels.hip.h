// rt_kernels.hip.h — gfx950 device code for the render hot path.
//
//   trace_ray     Scene::hit                 src/main.rs:776-814  (+ intersects_triangle 187-239,
//                                                                     intersects_sphere 241-262)
//   camera_ray    render's ray set-up        src/main.rs:995-1013
//   shade_hit     raytrace's hit branch      src/main.rs:829-969  (+ normal_at_hit 515-562,
//                                             get_tex_uv_coords 471-492, color_at 392-411,
//                                             normal_at 341-381, refract 320-327, reflectance 329-333)
//   sky           interp_sky                 src/main.rs:302-312
//
// Arithmetic contract: every value that decides where a ray goes (hit tests, normals, scatter
// directions, texture addresses) is computed in binary64 with the reference's operation order
// and without fused multiply-add (this file must be compiled with -ffp-contract=off; the only
// fused operations are the explicit __builtin_fma calls in pow6, which feed colour only).
// The f32 arithmetic is confined to box culling, which is conservative by construction.
#ifndef RT_KERNELS_HIP_H
#define RT_KERNELS_HIP_H

#include <hip/hip_runtime.h>

#include "rt_types.h"

#define RT_WAVE 64
#ifndef RT_BLOCK
#define RT_BLOCK 64
#endif

namespace rtk {

struct d3 {
  double x, y, z;
};
__device__ __forceinline__ d3 mk(double x, double y, double z) { return d3{x, y, z}; }
__device__ __forceinline__ d3 operator-(d3 a, d3 b) { return d3{a.x - b.x, a.y - b.y, a.z - b.z}; }
__device__ __forceinline__ d3 operator+(d3 a, d3 b) { return d3{a.x + b.x, a.y + b.y, a.z + b.z}; }
__device__ __forceinline__ d3 operator*(double s, d3 r) { return d3{s * r.x, s * r.y, s * r.z}; }
// src/main.rs:69-71: (a0*b0 + a1*b1) + a2*b2
__device__ __forceinline__ double dot(d3 a, d3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
__device__ __forceinline__ double length(d3 a) { return sqrt(a.x * a.x + a.y * a.y + a.z * a.z); }
__device__ __forceinline__ d3 cross(d3 a, d3 b) {
  return d3{a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
}

// Three divisions by one denominator, bit for bit what `a.x / l, a.y / l, a.z / l` give.
// A binary64 division is expanded by the compiler into: scale numerator and denominator
// (v_div_scale_f64), v_rcp_f64 of the denominator refined by two Newton steps, q = n * r, one
// correction of q by its remainder (v_div_fmas_f64), v_div_fixup_f64 for zeros, infinities and
// NaNs. The refined reciprocal depends on the denominator alone, and whenever nothing needs
// scaling the scale instructions return their inputs and v_div_fmas_f64 is a plain fused
// multiply-add; so one reciprocal and the remaining four steps per numerator perform the very
// same operations on the very same values. Nothing needs scaling when the denominator is in
// [2^-400, 2^400) and every numerator is zero or of magnitude in [2^-900, 2^500): quotient and
// remainder then stay in the normal range. div3_in_range assumes that; div3_by_length, for a
// denominator that no numerator exceeds by more than a rounding error (a vector and its length),
// checks the rest per lane and sends the lanes outside to the plain divisions.
__device__ __forceinline__ double refined_rcp(double l) {
  double r = __builtin_amdgcn_rcp(l);
  double e = __builtin_fma(-l, r, 1.0);
  r = __builtin_fma(r, e, r);
  e = __builtin_fma(-l, r, 1.0);
  r = __builtin_fma(r, e, r);
  return r;
}
__device__ __forceinline__ double div_by_rcp(double a, double l, double r) {
  const double q = a * r;
  const double rem = __builtin_fma(-l, q, a);
  return __builtin_amdgcn_div_fixup(__builtin_fma(rem, r, q), l, a);
}
__device__ __forceinline__ d3 div3_in_range(d3 a, double l) {
  const double r = refined_rcp(l);
  return d3{div_by_rcp(a.x, l, r), div_by_rcp(a.y, l, r), div_by_rcp(a.z, l, r)};
}
__device__ __forceinline__ d3 div3_by_length(d3 a, double l) {
  // binary exponents (0 for a zero, the true exponent for a denormal)
  const int ex = __builtin_amdgcn_frexp_exp(a.x), ey = __builtin_amdgcn_frexp_exp(a.y), ez = __builtin_amdgcn_frexp_exp(a.z);
  // high word of l minus that of 2^-400, against the span up to 2^400: false for infinities and
  // NaNs too (an infinite or NaN component makes the length one)
  const bool in_range = ((uint32_t)__double2hiint(l) - 0x26f00000u) < 0x32000000u && min(min(ex, ey), ez) >= -900;
  if (in_range) return div3_in_range(a, l);
  return d3{a.x / l, a.y / l, a.z / l};
}
// src/main.rs:87-90: every component divided by the length.
__device__ __forceinline__ d3 normalize(d3 a) { return div3_by_length(a, length(a)); }

// ------------------------------------------------------------------------------------------
// counter-based generator: draw = mix64(key + GOLDEN * ((stage << 20) + index + 1)), 52 bits
// ------------------------------------------------------------------------------------------
#define RT_GOLDEN 0x9e3779b97f4a7c15ull
#define RT_JITTER_STREAM (1u << 19) /* first draw index of a shading stage's second stream (a stage has 2^20 indices) */

__device__ __forceinline__ uint64_t mix64(uint64_t z) {
  z ^= z >> 30;
  z *= 0xbf58476d1ce4e5b9ull;
  z ^= z >> 27;
  z *= 0x94d049bb133111ebull;
  z ^= z >> 31;
  return z;
}
// key = mix64(mix64(mix64(seed + G) + G * (pixel + 1)) + G * (sample + 1)); the innermost term
// depends on the seed alone and comes from the host (seed_key).
__host__ __device__ __forceinline__ uint64_t seed_key(uint64_t seed) {
  uint64_t z = seed + RT_GOLDEN;
  z ^= z >> 30;
  z *= 0xbf58476d1ce4e5b9ull;
  z ^= z >> 27;
  z *= 0x94d049bb133111ebull;
  z ^= z >> 31;
  return z;
}
__device__ __forceinline__ uint64_t path_key(uint64_t key0, uint64_t pixel, uint64_t sample) {
  uint64_t k = mix64(key0 + RT_GOLDEN * (pixel + 1));
  k = mix64(k + RT_GOLDEN * (sample + 1));
  return k;
}
struct Rng {
  uint64_t x;  // key + GOLDEN * ((stage << 20) + index): advanced by GOLDEN per draw
  __device__ __forceinline__ void start(uint64_t key, uint32_t stage) { x = key + RT_GOLDEN * ((uint64_t)stage << 20); }
  __device__ __forceinline__ double u01() {
    x += RT_GOLDEN;
    // 52 random bits as the mantissa of a double in [1,2), minus 1 (what rand's f64 sampling does)
    return __longlong_as_double((long long)(0x3ff0000000000000ull | (mix64(x) >> 12))) - 1.0;
  }
  __device__ __forceinline__ double range(double lo, double hi) { return lo + (hi - lo) * u01(); }
  // Two uniforms on [-1,1) from ONE draw: -1 + 2 * (h / 2^32) for its high and for its low 32 bits
  // (resolution 2^-31). Used for the Marsaglia candidate pairs.
  __device__ __forceinline__ void pair_pm1(double& a, double& b) {
    x += RT_GOLDEN;
    const uint64_t z = mix64(x);
    // -1 + 2 * (h / 2^32) for a 32-bit half h is (h - 2^31) / 2^31, a multiple of 2^-31 below 1 in
    // magnitude: exact in binary64 however it is evaluated, so it is evaluated the cheap way
    a = (double)(int32_t)((uint32_t)(z >> 32) ^ 0x80000000u) * 4.656612873077392578125e-10;  // 2^-31
    b = (double)(int32_t)((uint32_t)z ^ 0x80000000u) * 4.656612873077392578125e-10;
  }
  __device__ __forceinline__ void skip(uint32_t n) { x += RT_GOLDEN * (uint64_t)n; }
};

// The point of the unit sphere an accepted candidate pair stands for, src/main.rs:272-277.
__device__ __forceinline__ d3 marsaglia_point(double x1, double x2) {
  const double sq = sqrt(1.0 - x1 * x1 - x2 * x2);
  return mk(2.0 * x1 * sq, 2.0 * x2 * sq, 1.0 - 2.0 * (x1 * x1 + x2 * x2));
}

// random_unit_sphere_surface, src/main.rs:264-279 (Marsaglia), capped at RT_REJECT_CAP tries.
__device__ __forceinline__ d3 random_unit_sphere_surface(Rng& r) {
  for (int tries = 0; tries < RT_REJECT_CAP; ++tries) {
    double x1, x2;
    r.pair_pm1(x1, x2);
    if (x1 * x1 + x2 * x2 >= 1.0) continue;
    double sq = sqrt(1.0 - x1 * x1 - x2 * x2);
    double x = 2.0 * x1 * sq;
    double y = 2.0 * x2 * sq;
    double z = 1.0 - 2.0 * (x1 * x1 + x2 * x2);
    return mk(x, y, z);
  }
  return mk(0.0, 0.0, 1.0);
}

// ------------------------------------------------------------------------------------------
// scene view handed to kernels by value
// ------------------------------------------------------------------------------------------
struct SceneK {
  const BvhNode* nodes;
  const BvhNodeQ* qnodes;  // the same tree on the 16-bit grid
  float q_lo[3], q_step[3];
  const TriGeom* geom;
  const TriShade* shade;
  const SphereRec* spheres;
  const rt_material* mats;
  const TextureRec* tex;
  int32_t root_ref;
  uint32_t n_spheres;
  uint32_t has_emission;  // some material has a non-zero emission
  int32_t leave_q;        // traversal: see trav_run
  double sky_light[3];    // normalize((1, 2, 1)), src/main.rs:307
  unsigned long long sph_always;  // bit i: sphere i (< 64) has no place in the tree and is tested against every ray
  uint32_t sph_places;            // some sphere has a place in the tree
  uint32_t deep_stack;            // host side: the tree is deep enough for the kernels' DEEP form (trav_run)
};

struct HitK {
  double t, u, v;
  int32_t slot;    // leaf-order triangle slot, or -1
  int32_t sphere;  // sphere record index, or -1
};

// Diagnostic build only (-DRT_STAMPS): per-lane cycle accounting by code section. RT_ACC(wc, s)
// charges the cycles since the previous stamp to section s. Never compiled into the library.
#ifdef RT_STAMPS
#define RT_NSEC 8
__shared__ unsigned long long s_stamp[RT_BLOCK / 64][RT_NSEC + 2];  // per wave: sections, t0
// Wave-level accounting: whichever lane is first among the active ones charges the cycles since
// the wave's previous stamp to section s.
#define RT_ACC(wc, s)                                                   \
  do {                                                                  \
    const unsigned long long m_ = __ballot(1);                          \
    if ((int)(threadIdx.x & 63) == __ffsll((long long)m_) - 1) {        \
      unsigned long long* w_ = s_stamp[threadIdx.x >> 6];               \
      const unsigned long long now_ = clock64();                        \
      w_[s] += now_ - w_[RT_NSEC];                                      \
      w_[RT_NSEC] = now_;                                               \
    }                                                                   \
  } while (0)
#else
#define RT_ACC(wc, s) \
  do {                \
  } while (0)
#endif

struct WorkCount {
  uint32_t rays, nodes, tris, spheres, shaded;
  // loop iterations as the WAVE executes them (counted by the first active lane of each
  // iteration): the divergence cost of the per-lane loops
  uint32_t wi_inner, wi_leaf, wi_reject;
};

__device__ __forceinline__ bool first_active_lane() {
  const unsigned long long m = __ballot(1);
  return (int)(threadIdx.x & 63) == __ffsll((long long)m) - 1;
}

// ------------------------------------------------------------------------------------------
// Scene::hit
// ------------------------------------------------------------------------------------------
// Box culling in f32 that can only err towards visiting: the ray origin is widened by `pad`
// (2^-21 of its magnitude: its f32 rounding, the rounding of o -+ pad, and the rounding of the
// product folded into the fused multiply-add each cost at most 2^-24), the far distance is
// scaled up by (1 + 2^-20) against the f32 rounding of the reciprocal direction and of the
// fused operation itself, and the pruning distance is the hit distance rounded up to f32 and
// scaled the same way; the boxes themselves were rounded outward and padded by the builder.
#define RT_K_FAR 1.00000095367431640625f /* 1 + 2^-20 */

__device__ __forceinline__ float f32_up(double x) {
  return __double2float_ru(x);
}

// Two binary32 values in a register pair, for the packed operations of the slab tests.
typedef float f2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f2 mk2(float a, float b) {
  f2 r;
  r.x = a;
  r.y = b;
  return r;
}
__device__ __forceinline__ f2 pk_fma(f2 a, f2 b, f2 c) { return __builtin_elementwise_fma(a, b, c); }

// Per-lane traversal state (the stack lives in LDS, one column per lane).
struct Trav {
  d3 o, d;
  float ix, iy, iz;                                 // reciprocal direction (grid nodes: grid step * reciprocal direction)
  float mx_lo, mx_hi, my_lo, my_hi, mz_lo, mz_hi;   // -(o -+ pad) * reciprocal direction
  float best32k;                                    // pruning distance for the f32 box tests
  double best_t, best_u, best_v;
  int32_t best_slot;
  int32_t cur;   // next node/leaf ref, RT_REF_SENTINEL when the traversal has finished
  uint32_t sel;                                     // grid nodes: per axis (one byte each) the rotation, 0 or 16 bits, that puts the nearer plane into the low half (trav_set_ray)
  int32_t* sp;   // top of the stack column (entries are `stride` ints apart)
  int32_t* base; // bottom of the stack column
  int32_t steps; // interior nodes this ray has visited
  unsigned long long spheres;  // bit i: the ray reached the place of sphere i in the tree
};

// Culling ray: t = b * inv - (o -+ pad) * inv, one fused operation per plane (culling only; the
// fused rounding perturbs the padded origin by less than the pad already allows for).
template <bool QUANT>
__device__ __forceinline__ void trav_set_ray(Trav& t, const SceneK& sc, const d3 o, const d3 d) {
  t.o = o;
  t.d = d;
  const double omax = fmax(fabs(o.x), fmax(fabs(o.y), fabs(o.z)));
  const float pad = f32_up(omax * 4.76837158203125e-07);  // 2^-21
  const float ox = (float)o.x, oy = (float)o.y, oz = (float)o.z;
  double dx = d.x, dy = d.y, dz = d.z;
  if (fabs(dx) < 1e-30) dx = signbit(dx) ? -1e-30 : 1e-30;
  if (fabs(dy) < 1e-30) dy = signbit(dy) ? -1e-30 : 1e-30;
  if (fabs(dz) < 1e-30) dz = signbit(dz) ? -1e-30 : 1e-30;
  // v_rcp_f32 of the rounded component: within 1.5 * 2^-23 of 1/d, which the far-plane factor
  // covers twice over ((1+e)/(1-e) with e = 1.5 * 2^-23 plus the fused operations' 2^-24 each is
  // 1 + 4.8e-7; the factor is 1 + 9.5e-7). Three binary64 divisions per ray cost ten times more.
  t.ix = __builtin_amdgcn_rcpf((float)dx);
  t.iy = __builtin_amdgcn_rcpf((float)dy);
  t.iz = __builtin_amdgcn_rcpf((float)dz);
  // for a box's min plane use o + pad, for its max plane o - pad (pads the box, whatever the sign)
  t.mx_lo = -(ox - pad) * t.ix;
  t.mx_hi = -(ox + pad) * t.ix;
  t.my_lo = -(oy - pad) * t.iy;
  t.my_hi = -(oy + pad) * t.iy;
  t.mz_lo = -(oz - pad) * t.iz;
  t.mz_hi = -(oz + pad) * t.iz;
  if (QUANT) {
    // plane = lo + q * step, so t = q * (step * inv) + (lo * inv - (o -+ pad) * inv). The three
    // extra roundings each move a plane distance by at most 2^-24 of (scene extent * |inv|); the
    // grid planes lie a whole grid step (2^-16 of the extent) outside the binary32 planes.
    t.mx_lo += sc.q_lo[0] * t.ix;
    t.mx_hi += sc.q_lo[0] * t.ix;
    t.my_lo += sc.q_lo[1] * t.iy;
    t.my_hi += sc.q_lo[1] * t.iy;
    t.mz_lo += sc.q_lo[2] * t.iz;
    t.mz_hi += sc.q_lo[2] * t.iz;
    t.ix *= sc.q_step[0];
    t.iy *= sc.q_step[1];
    t.iz *= sc.q_step[2];
    // A grid node keeps the two planes of a child on an axis in the halves of one word, min below
    // max. Which of them the ray meets first is the sign of its direction on that axis, the same at
    // every node: a rotation by 0 or 16 bits per axis swaps the halves for a negative direction, so that the
    // low half always holds the nearer plane, and the padded origin terms are swapped with them.
    // The distances are those the unordered form takes the smaller and the larger of (the padded
    // box has its min plane below its max plane), without the twelve min/max per visit.
    t.sel = (t.ix < 0.0f ? 16u : 0u) | (t.iy < 0.0f ? 16u << 8 : 0u) | (t.iz < 0.0f ? 16u << 16 : 0u);
    // after this, m*_hi goes with the low half (the nearer plane), m*_lo with the high half
    if (t.ix < 0.0f) { const float a = t.mx_lo; t.mx_lo = t.mx_hi; t.mx_hi = a; }
    if (t.iy < 0.0f) { const float a = t.my_lo; t.my_lo = t.my_hi; t.my_hi = a; }
    if (t.iz < 0.0f) { const float a = t.mz_lo; t.mz_lo = t.mz_hi; t.mz_hi = a; }
  }
}

template <bool QUANT = false>
__device__ __forceinline__ void trav_begin(Trav& t, const SceneK& sc, const d3 o, const d3 d, int32_t* stk) {
  trav_set_ray<QUANT>(t, sc, o, d);
  t.best32k = __int_as_float(0x7f800000);
  t.best_t = __longlong_as_double(0x7ff0000000000000ll);
  t.best_u = 0.0;
  t.best_v = 0.0;
  t.best_slot = -1;
  t.sp = stk;
  t.base = stk;
  *t.sp = RT_REF_SENTINEL;
  t.cur = sc.root_ref;
  t.steps = 0;
  t.spheres = 0ull;
}

// Runs the traversal to its end (t.cur == RT_REF_SENTINEL).
// DEEP is the kernels' general form: it is also the one that knows about spheres with places in
// the tree (scenes of nine spheres or more), so that the form every other scene runs carries neither
// (1-1.5 % of a frame).
// The stack is a full stack in LDS, sized from the tree's depth: up to 39 entries (10 KB a wave)
// that costs a kernel of five waves per SIMD nothing. DEEP, for the trees beyond (the builders stop
// at RT_BVH_MAX_DEPTH; only bottom-up clustering of a soup of overlapping triangles gets there):
// RT_DEEP_LDS entries in LDS and the deeper ones in private memory, which keeps the waves a deep
// stack would cost (measured on a soup's tree of depth 56: 2259 -> 1847 ms a frame) at the price
// of a comparison on every push and pop (4-6 % on the scenes of ordinary depth, hence not for them).
#define RT_DEEP_LDS 16
template <bool COUNT, bool QUANT = false, bool DEEP = false>
__device__ __forceinline__ void trav_run(Trav& t, const SceneK& sc, const int stride, WorkCount& wc) {
  const double eps = 0.00000001;  // src/main.rs:185
  const d3 o = t.o, d = t.d;
  const float ix = t.ix, iy = t.iy, iz = t.iz;
  const float mx_lo = t.mx_lo, mx_hi = t.mx_hi, my_lo = t.my_lo, my_hi = t.my_hi, mz_lo = t.mz_lo, mz_hi = t.mz_hi;
  const uint32_t sel = t.sel;
  float best32k = t.best32k;
  double best_t = t.best_t, best_u = t.best_u, best_v = t.best_v;
  int32_t best_slot = t.best_slot;
  int32_t cur = t.cur;
  int32_t* sp = t.sp;
  // DEEP: the first RT_DEEP_LDS stack entries in LDS, deeper ones in the lane's private memory.
  int32_t spill[DEEP ? RT_BVH_MAX_DEPTH + 2 - RT_DEEP_LDS : 1];
  const int32_t* const lds_end = DEEP ? t.base + RT_DEEP_LDS * stride : nullptr;
#define RT_PUSH(v)                                                   \
  do {                                                               \
    sp += stride;                                                    \
    if (!DEEP || sp < lds_end) *sp = (v);                            \
    else spill[(int)(sp - lds_end) / RT_BLOCK] = (v);                \
  } while (0)
#define RT_POP()                                                                       \
  do {                                                                                 \
    cur = (!DEEP || sp < lds_end) ? *sp : spill[(int)(sp - lds_end) / RT_BLOCK];       \
    sp -= stride;                                                                      \
  } while (0)
  int steps = 0;
  unsigned long long spheres = t.spheres;

  RT_ACC(wc, 2);
  while (cur != RT_REF_SENTINEL) {
    RT_ACC(wc, 5);
    // The lanes that have reached a leaf wait for the others to reach theirs, but not for a few
    // stragglers: once fewer than leave_q / 64 of the lanes still traversing are stepping through
    // nodes, those pause (cur stays >= 0) while the leaves found so far are tested. Only the
    // interleaving of the lanes changes, not what any ray visits.
    const int leave_at = (__popcll(__ballot(1)) * sc.leave_q) >> 6;  // the lanes inside this loop vote
    while (cur >= 0) {
      if ((int)__popcll(__ballot(1)) < leave_at) break;
      steps += 1;
      float4 bx, by, bz;
      int4 ch;
      if (QUANT) {
        // a 32-bit byte offset from the (uniform) base: one shift per lane instead of 64-bit address arithmetic
        const uint4* np = reinterpret_cast<const uint4*>(reinterpret_cast<const char*>(sc.qnodes) + ((uint32_t)cur << 5));
        uint4 w0 = np[0], w1 = np[1];
        // nearer plane into the low half of every word (see trav_set_ray); the rotation takes the low five bits of its operand
        const uint32_t sel_x = sel, sel_y = sel >> 8, sel_z = sel >> 16;
        w0.x = __builtin_amdgcn_alignbit(w0.x, w0.x, sel_x);
        w0.y = __builtin_amdgcn_alignbit(w0.y, w0.y, sel_x);
        w0.z = __builtin_amdgcn_alignbit(w0.z, w0.z, sel_y);
        w0.w = __builtin_amdgcn_alignbit(w0.w, w0.w, sel_y);
        w1.x = __builtin_amdgcn_alignbit(w1.x, w1.x, sel_z);
        w1.y = __builtin_amdgcn_alignbit(w1.y, w1.y, sel_z);
        bx = make_float4((float)(w0.x & 0xffffu), (float)(w0.x >> 16), (float)(w0.y & 0xffffu), (float)(w0.y >> 16));
        by = make_float4((float)(w0.z & 0xffffu), (float)(w0.z >> 16), (float)(w0.w & 0xffffu), (float)(w0.w >> 16));
        bz = make_float4((float)(w1.x & 0xffffu), (float)(w1.x >> 16), (float)(w1.y & 0xffffu), (float)(w1.y >> 16));
        ch = make_int4((int)w1.z, (int)w1.w, 0, 0);
      } else {
        const float4* np = reinterpret_cast<const float4*>(reinterpret_cast<const char*>(sc.nodes) + ((uint32_t)cur << 6));
        bx = np[0];
        by = np[1];
        bz = np[2];
        ch = reinterpret_cast<const int4*>(np)[3];
      }
      if (COUNT) {
        wc.nodes++;
        if (first_active_lane()) wc.wi_inner++;
      }
      // A NaN plane distance (inf - inf when a direction component is zero and the numbers are
      // huge) drops out of fminf/fmaxf, which then ignore that plane: still conservative.
      // Both planes of an axis in one packed operation (v_pk_fma_f32): {min, max} * inv + {m_hi, m_lo}.
      const f2 x0 = pk_fma(mk2(bx.x, bx.y), mk2(ix, ix), mk2(mx_hi, mx_lo));
      const f2 y0 = pk_fma(mk2(by.x, by.y), mk2(iy, iy), mk2(my_hi, my_lo));
      const f2 z0 = pk_fma(mk2(bz.x, bz.y), mk2(iz, iz), mk2(mz_hi, mz_lo));
      const f2 x1 = pk_fma(mk2(bx.z, bx.w), mk2(ix, ix), mk2(mx_hi, mx_lo));
      const f2 y1 = pk_fma(mk2(by.z, by.w), mk2(iy, iy), mk2(my_hi, my_lo));
      const f2 z1 = pk_fma(mk2(bz.z, bz.w), mk2(iz, iz), mk2(mz_hi, mz_lo));
      float tn0, tf0, tn1, tf1;
      if (QUANT) {
        // .x is the distance to the nearer plane of the axis, .y to the farther one
        tn0 = fmaxf(fmaxf(fmaxf(x0.x, y0.x), z0.x), 0.0f);
        tf0 = fminf(fminf(x0.y, y0.y), z0.y);
        tn1 = fmaxf(fmaxf(fmaxf(x1.x, y1.x), z1.x), 0.0f);
        tf1 = fminf(fminf(x1.y, y1.y), z1.y);
      } else {
        // child 0
        tn0 = fmaxf(fminf(x0.x, x0.y), fminf(y0.x, y0.y));
        tf0 = fminf(fmaxf(x0.x, x0.y), fmaxf(y0.x, y0.y));
        tn0 = fmaxf(fmaxf(tn0, fminf(z0.x, z0.y)), 0.0f);
        tf0 = fminf(tf0, fmaxf(z0.x, z0.y));
        // child 1
        tn1 = fmaxf(fminf(x1.x, x1.y), fminf(y1.x, y1.y));
        tf1 = fminf(fmaxf(x1.x, x1.y), fmaxf(y1.x, y1.y));
        tn1 = fmaxf(fmaxf(tn1, fminf(z1.x, z1.y)), 0.0f);
        tf1 = fminf(tf1, fmaxf(z1.x, z1.y));
      }
      const f2 tf = mk2(tf0, tf1) * mk2(RT_K_FAR, RT_K_FAR);
      tf0 = tf.x;
      tf1 = tf.y;

      const bool h0 = tn0 <= fminf(tf0, best32k);
      const bool h1 = tn1 <= fminf(tf1, best32k);
      if (h0 && h1) {
        const bool swap = tn1 < tn0;
        const int32_t near = swap ? ch.y : ch.x;
        const int32_t far = swap ? ch.x : ch.y;
        RT_PUSH(far);
        cur = near;
      } else if (h0) {
        cur = ch.x;
      } else if (h1) {
        cur = ch.y;
      } else {
        RT_POP();
      }
    }
    RT_ACC(wc, 4);
    if (cur == RT_REF_SENTINEL) break;
    if (cur >= 0) continue;  // paused on the way to its next leaf
    // leaf
    const uint32_t leaf = (uint32_t)~cur;
    const uint32_t first = leaf >> RT_LEAF_BITS, cnt = (leaf & (uint32_t)(RT_LEAF_MAX - 1)) + 1u;
    for (uint32_t i = 0; i < cnt; ++i) {
      if (COUNT && first_active_lane()) wc.wi_leaf++;
      const TriGeom* g = reinterpret_cast<const TriGeom*>(reinterpret_cast<const char*>(sc.geom) + (first + i) * (uint32_t)sizeof(TriGeom));
      const d3 a = mk(g->a[0], g->a[1], g->a[2]);
      const d3 a_to_b = mk(g->ab[0], g->ab[1], g->ab[2]);
      const d3 a_to_c = mk(g->ac[0], g->ac[1], g->ac[2]);
      const int32_t flat = g->flat;
      if (COUNT) wc.tris++;
      // a sphere's place in the tree (all-zero geometry: the test below fails by itself): note the sphere
      if (DEEP && sc.sph_places && flat < 0) spheres |= 1ull << ((uint32_t)~flat & 63u);
      // intersects_triangle, src/main.rs:187-239
      const d3 u_vec = cross(d, a_to_c);
      const double det = dot(a_to_b, u_vec);
      bool ok = !(det < eps && det > -eps);
      const double inv_det = 1.0 / det;
      const d3 a_to_origin = o - a;
      const double u = dot(a_to_origin, u_vec) * inv_det;
      ok = ok && (u >= 0.0 && u <= 1.0);
      const d3 v_vec = cross(a_to_origin, a_to_b);
      const double v = dot(d, v_vec) * inv_det;
      ok = ok && !(v < 0.0 || u + v > 1.0);
      const double dist = dot(a_to_c, v_vec) * inv_det;
      ok = ok && (dist > eps);
      // src/main.rs:791: strict <, so of equal distances the first in flat order wins. The flat index
      // of the best hit so far is read again from its record in the rare case of a tie (a register
      // fewer where the kernels hold most).
      bool better = ok && dist < best_t;
      if (ok && dist == best_t) {
        const int32_t best_flat = reinterpret_cast<const TriGeom*>(reinterpret_cast<const char*>(sc.geom) + (uint32_t)best_slot * (uint32_t)sizeof(TriGeom))->flat;
        better = flat < best_flat;
      }
      if (better) {
        best_t = dist;
        best_u = u;
        best_v = v;
        best_slot = (int32_t)(first + i);
        best32k = f32_up(dist) * RT_K_FAR;
      }
    }
    RT_POP();
  }
  RT_ACC(wc, 5);
#undef RT_PUSH
#undef RT_POP
  t.best32k = best32k;
  t.best_t = best_t;
  t.best_u = best_u;
  t.best_v = best_v;
  t.best_slot = best_slot;
  t.cur = cur;
  t.sp = sp;
  t.steps = steps;
  t.spheres = spheres;
}

// intersects_sphere, src/main.rs:241-262, for sphere i, and Scene::hit's "strictly nearer replaces"
// (src/main.rs:805).
__device__ __forceinline__ void test_sphere(const SceneK& sc, uint32_t i, const d3 o, const d3 d, HitK& h) {
  const double eps = 0.00000001;
  const SphereRec* s = sc.spheres + i;
  const d3 oc = o - mk(s->c[0], s->c[1], s->c[2]);
  const double a = dot(d, d);
  const double half_b = dot(oc, d);
  const double c = dot(oc, oc) - s->r * s->r;
  const double discriminant = half_b * half_b - a * c;
  if (discriminant < 0.0) return;
  const double sqrtd = sqrt(discriminant);
  double root = (-half_b - sqrtd) / a;
  if (root < eps) {
    root = (-half_b + sqrtd) / a;
    if (root < eps) return;
  }
  // strictly nearer replaces, so of two spheres at the same distance the one tested first — the
  // lower index — wins in the reference; written so that it wins here in whatever order they are tested
  if (root < h.t || (root == h.t && h.sphere > (int32_t)i)) {
    h.t = root;
    h.slot = -1;
    h.sphere = (int32_t)i;
  }
}

// The sphere loop of Scene::hit, src/main.rs:802-810, on top of the nearest mesh hit: spheres in
// object order, strictly nearer replaces. The reference runs the test, with its square root and
// one or two divisions, on every sphere for every ray. Here the tree culls spheres as it culls
// triangles: a sphere has a place in a leaf, found by its bounding box, and only the spheres whose
// place the ray reached (t.spheres) are tested — a ray cannot hit a sphere without entering its
// box. Spheres without a place (not finite, or beyond the 64 the mask has bits for:
// sc.sph_always and the indices from 64 on) are tested always.
// PLACES: the scene's spheres may have places in the tree (the kernels' DEEP form, which is also the
// form for scenes with such places: see trav_run); without it every sphere is tested against every ray.
template <bool COUNT, bool PLACES = false>
__device__ __forceinline__ void finish_hit(const SceneK& sc, const Trav& t, HitK& h, WorkCount& wc) {
  const d3 o = t.o, d = t.d;
  h.t = t.best_t;
  h.u = t.best_u;
  h.v = t.best_v;
  h.slot = t.best_slot;
  h.sphere = -1;
  // (a) the spheres every ray tests: the same index in every lane, so the record is read once per
  // wave into scalar registers
  const uint32_t n_low = PLACES ? min(sc.n_spheres, (uint32_t)RT_TREE_SPHERES) : sc.n_spheres;
  for (uint32_t i = 0; i < n_low; ++i) {
    if (PLACES && !((sc.sph_always >> i) & 1ull)) continue;
    if (COUNT) wc.spheres++;
    test_sphere(sc, i, o, d, h);
  }
  if (!PLACES) {
    RT_ACC(wc, 6);
    return;
  }
  // (b) the spheres whose place this ray reached
  unsigned long long mask = sc.sph_places ? t.spheres : 0ull;
  while (mask) {
    const uint32_t i = (uint32_t)__ffsll((long long)mask) - 1u;
    mask &= mask - 1ull;
    if (COUNT) wc.spheres++;
    test_sphere(sc, i, o, d, h);
  }
  for (uint32_t i = (uint32_t)RT_TREE_SPHERES; i < sc.n_spheres; ++i) {
    if (COUNT) wc.spheres++;
    test_sphere(sc, i, o, d, h);
  }
  RT_ACC(wc, 6);
}

template <bool COUNT, bool DEEP = false>
__device__ __forceinline__ void trace_ray(const SceneK& sc, const d3 o, const d3 d, int32_t* stk, const int stride,
                                          HitK& h, WorkCount& wc) {
  Trav t;
  trav_begin(t, sc, o, d, stk);
  trav_run<COUNT, false, DEEP>(t, sc, stride, wc);
  finish_hit<COUNT, DEEP>(sc, t, h, wc);
}

// ------------------------------------------------------------------------------------------
// sky — interp_sky, src/main.rs:302-312
// ------------------------------------------------------------------------------------------
// x^6 in double-double arithmetic, rounded once: agrees with a correctly rounded pow(x, 6.0)
// except when the exact value lies within 2^-100 of a rounding boundary. Colour only.
__device__ __forceinline__ double pow6(double x) {
  const double p2 = x * x;
  const double e2 = __builtin_fma(x, x, -p2);
  double p3 = p2 * x;
  double e3 = __builtin_fma(p2, x, -p3) + e2 * x;
  const double s3 = p3 + e3;
  e3 = e3 - (s3 - p3);
  p3 = s3;
  const double p6 = p3 * p3;
  const double e6 = __builtin_fma(p3, p3, -p6) + 2.0 * (p3 * e3);
  return p6 + e6;
}

__device__ __forceinline__ d3 sky(const SceneK& sc, const d3 dir) {
  const d3 c1 = mk(0.4, 0.6, 0.8);
  const d3 c2 = mk(0.8, 0.8, 0.8);
  const d3 light_c = mk(1.5, 1.5, 1.5);
  // normalize((1, 2, 1)), src/main.rs:307: the same three correctly rounded quotients whoever
  // computes them, so the host does, once (sky_light_dir)
  const d3 light_dir = mk(sc.sky_light[0], sc.sky_light[1], sc.sky_light[2]);
  double t = dot(dir, light_dir);
  t = pow6((t + 1.0) / 2.0);
  return (1.0 - t) * (dir.y * c2 + (1.0 - dir.y) * c1) + t * light_c;
}

// ------------------------------------------------------------------------------------------
// textures
// ------------------------------------------------------------------------------------------
// `(x).floor() as u32` saturates (negative and NaN -> 0); image::get_pixel would panic one
// past the edge (exact u == 1 or v == 0 only), here the address is clamped to the last texel.
__device__ __forceinline__ uint32_t texel_index(double x, uint32_t n) {
  const double f = floor(x);
  uint32_t i = 0;
  if (f > 0.0) i = f >= 4294967295.0 ? 4294967295u : (uint32_t)f;
  return i >= n ? n - 1 : i;
}
__device__ __forceinline__ uchar4 fetch_texel(const TextureRec& t, double u, double v) {
  const uint32_t x = texel_index(u * (double)t.w, t.w);
  const uint32_t y = texel_index((1.0 - v) * (double)t.h, t.h);
  return *reinterpret_cast<const uchar4*>(t.rgba + rt_texel_offset(x, y, t.blocks_x) * 4u);
}

__device__ __forceinline__ double bary(double a, double b, double c, double u, double v) {  // src/main.rs:741-745
  return (1.0 - u - v) * a + u * b + v * c;
}

// f64::powi(5): square-and-multiply from the low bit
__device__ __forceinline__ double powi5(double a) {
  const double a2 = a * a;
  const double a4 = a2 * a2;
  return a * a4;
}

// ------------------------------------------------------------------------------------------
// raytrace's hit branch (a hit never ends a path: it continues until a miss or the depth cap):
// fills the next ray and updates throughput and emitted radiance.
// ------------------------------------------------------------------------------------------
struct ShadeIn {
  d3 hit_pos;      // origin + distance * dir, src/main.rs:831
  double u, v;     // barycentrics (mesh hits)
  int32_t slot;    // leaf-order triangle slot, or -1
  int32_t sphere;  // sphere record index, or -1
};

// EMIT = false: the scene has no emitter, so the emitted radiance a path carries stays what it was
// (zero) and costs no registers.
template <bool COUNT, bool EMIT = true>
__device__ __forceinline__ void shade_hit(const SceneK& sc, const ShadeIn& h, const uint64_t key, const int depth, d3& o,
                                          d3& dir, d3& thr, d3& rad, WorkCount& wc) {
  const double eps = 0.00000001;
  const d3 hit_pos = h.hit_pos;
  d3 normal;
  double tex_u = 0.0, tex_v = 1.0;  // src/main.rs:490
  int32_t obj;
  if (h.slot >= 0) {
    const TriShade* s = sc.shade + h.slot;
    obj = s->obj;
    const d3 n0 = mk(s->n[0][0], s->n[0][1], s->n[0][2]), n1 = mk(s->n[1][0], s->n[1][1], s->n[1][2]),
             n2 = mk(s->n[2][0], s->n[2][1], s->n[2][2]);
    double uv[3][2];
    for (int k = 0; k < 3; ++k) {
      uv[k][0] = s->uv[k][0];
      uv[k][1] = s->uv[k][1];
    }
    // src/main.rs:526-531 — bary_interp of the vertex normals, not normalised
    const double w = 1.0 - h.u - h.v;
    normal = (w * n0 + h.u * n1) + h.v * n2;
    tex_u = bary(uv[0][0], uv[1][0], uv[2][0], h.u, h.v);  // src/main.rs:485-486
    tex_v = bary(uv[0][1], uv[1][1], uv[2][1], h.u, h.v);
  } else {
    const SphereRec* s = sc.spheres + h.sphere;
    obj = s->obj;
    normal = normalize(hit_pos - mk(s->c[0], s->c[1], s->c[2]));  // src/main.rs:556-557
  }
  const rt_material* m = sc.mats + obj;
  const int kind = m->kind;
  Rng rng;
  rng.start(key, (uint32_t)depth + 1u);

  if (kind == RT_MAT_OPAQUE) {
    if (h.slot >= 0 && m->normal_tex >= 0) {
      // NormalKind::normal_at, src/main.rs:347-378
      const uchar4 px = fetch_texel(sc.tex[m->normal_tex], tex_u, tex_v);
      const d3 ts_normal = normalize(div3_in_range(mk((double)px.x, (double)px.y, (double)px.z), 255.0));  // 0..255 over 255
      d3 t = cross(normal, mk(0.0, 1.0, 0.0));
      if (fabs(length(t)) < eps) t = cross(normal, mk(0.0, 0.0, 1.0));
      t = normalize(t);
      const d3 b = cross(normal, t);
      const d3 map_n = 2.0 * ts_normal - mk(1.0, 1.0, 1.0);
      // tbn * (map_n, 1) row by row (four-term sums, src/main.rs:160-165), then divide by w
      const double w4 = 0.0 * map_n.x + 0.0 * map_n.y + 0.0 * map_n.z + 1.0 * 1.0;
      const d3 nn = mk((t.x * map_n.x + b.x * map_n.y + normal.x * map_n.z + 0.0 * 1.0) / w4,
                       (t.y * map_n.x + b.y * map_n.y + normal.y * map_n.z + 0.0 * 1.0) / w4,
                       (t.z * map_n.x + b.z * map_n.y + normal.z * map_n.z + 0.0 * 1.0) / w4);
      normal = normalize(nn);
    }
    // ColorKind::color_at, src/main.rs:392-411
    d3 point;
    if (m->color_tex >= 0) {
      const uchar4 px = fetch_texel(sc.tex[m->color_tex], tex_u, tex_v);
      point = div3_in_range(mk((double)px.x, (double)px.y, (double)px.z), 255.0);  // 0..255 over 255
    } else {
      point = mk(m->color[0], m->color[1], m->color[2]);
    }
    const double metallic = m->metallic, diffuse = m->diffuse, how_metallic = m->how_metallic;
    // src/main.rs:925-929 unrolled: L = albedo * (L_in + emission)
    thr = mk(thr.x * point.x, thr.y * point.y, thr.z * point.z);
    if (EMIT) rad = rad + mk(thr.x * m->emission[0], thr.y * m->emission[1], thr.z * m->emission[2]);
    // src/main.rs:869-883. The reference draws a Marsaglia point for the metal jitter (rejection
    // loop 265-278), three unused uniforms (871-873), then Marsaglia points until one lies in the
    // normal's hemisphere (876-883). Here the hemisphere loop only DECIDES which candidate pair it
    // takes, and the point is built afterwards, once, from the chosen pair.
    RT_ACC(wc, 0);
    // The hemisphere sample takes its uniforms from the stage's draws 0, 1, ...; the jitter and the
    // three unused draws take theirs from a stream of their own (RT_JITTER_STREAM), so that a
    // material whose jitter never reaches the result (below) does not draw it at all.
    // The hemisphere sample: one loop over candidate pairs (nested loops would cost a wave the
    // maximum trip count of every level): `inner` counts the pairs of the current Marsaglia call,
    // `outer` the calls. Whether a point lies in the hemisphere is the sign of dot(point, normal) as
    // the reference rounds it; the loop first forms that dot product with the hardware's
    // approximate square root (v_sqrt_f64, 2^-23 relative) and fused operations, which is off by
    // less than 2^-21 of |n.x| + |n.y| + |n.z|; only a value within 2^-20 of that sum of zero —
    // about one candidate in a million — is settled by the reference's own arithmetic.
    double x1 = 0.0, x2 = 0.0;
    int b_kind = 0;  // 0: no point within the cap, 1: the point of (x1, x2), 2: a call's fallback point (0, 0, 1)
    {
      const double doubt = (fabs(normal.x) + fabs(normal.y) + fabs(normal.z)) * 9.5367431640625e-07;  // 2^-20
      int inner = 0, outer = 0;
      for (;;) {
        if (COUNT && first_active_lane()) wc.wi_reject++;
        rng.pair_pm1(x1, x2);
        inner += 1;
        const double r2 = x1 * x1 + x2 * x2;
        const bool accept = !(r2 >= 1.0);
        if (!accept && inner < RT_REJECT_CAP) continue;
        inner = 0;
        outer += 1;
        bool inside;
        if (accept) {
          const double sq = __builtin_amdgcn_sqrt(1.0 - r2);
          const double approx = __builtin_fma(2.0 * sq, __builtin_fma(normal.x, x1, normal.y * x2), normal.z * __builtin_fma(-2.0, r2, 1.0));
          inside = approx > 0.0;
          if (!(fabs(approx) > doubt)) inside = dot(marsaglia_point(x1, x2), normal) > 0.0;  // also when the normal is not finite
        } else {
          inside = dot(mk(0.0, 0.0, 1.0), normal) > 0.0;
        }
        if (inside) {
          b_kind = accept ? 1 : 2;
          break;
        }
        if (outer >= RT_REJECT_CAP) break;
      }
    }
    RT_ACC(wc, 1);
    d3 diffuse_out = mk(0.0, 0.0, 0.0);
    if (b_kind == 1) diffuse_out = marsaglia_point(x1, x2);
    if (b_kind == 2) diffuse_out = mk(0.0, 0.0, 1.0);
    diffuse_out = diffuse * diffuse_out + normal;  // src/main.rs:884
    if (length(diffuse_out) < eps) {
      diffuse_out = normal;
    } else {
      diffuse_out = normalize(diffuse_out);
    }
    // src/main.rs:900: out = how_metallic * fuzzy + (1 - how_metallic) * diffuse_out. For a purely
    // diffuse material (how_metallic == 0) that is (+-0) + diffuse_out = diffuse_out, bit for bit,
    // provided the fuzzy mirror direction is finite and no component of diffuse_out is a zero (whose
    // sign the sum could change); then the mirror direction — the viewing direction, its reflection,
    // the jitter point and three normalisations — is not computed at all. It is finite when
    //   * the incoming direction can be normalised (its squared length neither under- nor overflows):
    //     c = normalize(-dir) is then a unit vector;
    //   * r = 2 (c.n) n - c can be normalised: |r|^2 = 1 + 4 (c.n)^2 (|n|^2 - 1) >= 1 - 4 |n|^2 (1 - |n|^2),
    //     which is at least 0.04 unless |n|^2 lies in [0.4, 0.6] (and r = -c when c.n < 0);
    //   * the jitter is shorter than the unit mirror direction it is added to: |metallic| <= 1/2.
    // Everything else takes the reference's full formula.
    const double dd = dot(dir, dir), nn = dot(normal, normal);
    const bool diffuse_only = how_metallic == 0.0 && fabs(metallic) <= 0.5 && dd > 1e-200 && dd < 1e200 && (nn < 0.4 || (nn > 0.6 && nn < 4.0)) &&
                              diffuse_out.x != 0.0 && diffuse_out.y != 0.0 && diffuse_out.z != 0.0;
    d3 out_dir = diffuse_out;
    if (!diffuse_only) {
      const d3 camera_dir = normalize(-1.0 * dir);  // src/main.rs:835
      // src/main.rs:851-857
      const double cdn = dot(camera_dir, normal);
      d3 r;
      if (cdn < 0.0) {
        r = -1.0 * camera_dir;
      } else {
        r = 2.0 * cdn * normal - camera_dir;
      }
      const d3 reflected_camera = normalize(r);
      // the jitter's point, from the stage's second stream (not unrolled: four in five lanes are done
      // after the first pair, and a wave issues every instruction of an unrolled body for the one
      // lane that is not)
      Rng jrng;
      jrng.start(key, (uint32_t)depth + 1u);
      jrng.skip(RT_JITTER_STREAM);
      d3 pt = mk(0.0, 0.0, 1.0);  // fallback after RT_REJECT_CAP rejections
#pragma clang loop vectorize(disable) interleave(disable) unroll(disable)
      for (int tries = 0; tries < RT_REJECT_CAP; ++tries) {
        if (COUNT && first_active_lane()) wc.wi_reject++;
        double j1, j2;
        jrng.pair_pm1(j1, j2);
        if (!(j1 * j1 + j2 * j2 >= 1.0)) {
          pt = marsaglia_point(j1, j2);
          break;
        }
      }
      const d3 jit = metallic * pt;                                                 // src/main.rs:869
      const d3 fuzzy_out = normalize(jit + reflected_camera);                       // src/main.rs:896
      out_dir = how_metallic * fuzzy_out + (1.0 - how_metallic) * diffuse_out;     // src/main.rs:900
    }
    o = hit_pos + eps * out_dir;                                                        // src/main.rs:903
    dir = out_dir;
  } else {
    // Dielectric, src/main.rs:936-969
    const double ior = m->ior;
    const d3 camera_dir = normalize(-1.0 * dir);  // src/main.rs:835
    const bool front_face = dot(normal, camera_dir) > 0.0;
    const d3 n = front_face ? normal : -1.0 * normal;
    const double eta = front_face ? 1.0 / ior : ior;
    const double cos_theta = fmin(dot(camera_dir, n), 1.0);
    const double sin_theta = sqrt(1.0 - cos_theta * cos_theta);
    const bool cannot_refract = eta * sin_theta > 1.0;
    // reflectance, src/main.rs:329-333
    double r0 = (1.0 - eta) / (1.0 + eta);
    r0 = r0 * r0;
    const double refl = r0 + (1.0 - r0) * powi5(1.0 - cos_theta);
    const double xi = rng.range(0.0, 1.0);
    d3 nd;
    if (cannot_refract || refl > xi) {
      nd = normalize(2.0 * dot(camera_dir, n) * n - camera_dir);
    } else {
      // refract, src/main.rs:320-327
      const d3 uv = -1.0 * camera_dir;
      const double ct = fmin(dot(-1.0 * uv, n), 1.0);
      const d3 r_out_perp = eta * (uv + ct * n);
      const d3 r_out_parallel = -sqrt(fabs(1.0 - dot(r_out_perp, r_out_perp))) * n;
      nd = normalize(r_out_perp + r_out_parallel);
    }
    o = hit_pos + eps * nd;  // src/main.rs:966
    dir = nd;
  }
}

// ------------------------------------------------------------------------------------------
// camera — src/main.rs:995-1013
// ------------------------------------------------------------------------------------------
struct CameraK {
  double tan_half;  // (fov_y / 2).tan(), computed on the host
  double aspect;    // width / height
  double dof;
  uint32_t width, height;
  uint32_t jitter;
};

__device__ __forceinline__ void camera_ray(const CameraK& cam, uint32_t u, uint32_t v, uint64_t key, d3& o, d3& dir) {
  Rng rng;
  rng.start(key, 0u);
  double x_offset = rng.range(-0.3, 0.3);
  double y_offset = rng.range(-0.3, 0.3);
  if (!cam.jitter) {
    x_offset = 0.0;
    y_offset = 0.0;
  }
  const double p_x = (2.0 * (((double)u + 0.5 + x_offset) / (double)cam.width) - 1.0) * cam.tan_half * cam.aspect;
  const double p_y = (1.0 - 2.0 * (((double)v + 0.5 + y_offset) / (double)cam.height)) * cam.tan_half;
  // src/main.rs:1007-1008. With DOF_STRENGTH = 0 (the reference's setting) both draws give
  // -0 + 0*u = +0 whatever u is, so the two generator calls are skipped: they are the camera
  // stage's last draws, and every later stage of the path starts its own counter.
  double ox = 0.0, oy = 0.0;
  if (cam.dof != 0.0) {
    ox = rng.range(-cam.dof, cam.dof);
    oy = rng.range(-cam.dof, cam.dof);
  }
  o = mk(ox, oy, 0.0);
  const d3 d = 2.4 * normalize(mk(p_x, p_y, -1.0));
  dir = normalize(d - o);
}

}  // namespace rtk

#endif  // RT_KERNELS_HIP_H
