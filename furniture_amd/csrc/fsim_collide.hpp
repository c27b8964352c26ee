// fsim_collide.hpp -- P3 collision for one env / one wave.
//
// lanes = colliding geoms (world poses), then lanes = candidate pairs (mask + bounding-sphere
// broadphase, wave-ballot compaction in pair order => deterministic contact order), then
// lanes = surviving pairs (primitive narrow phase).  Contact convention: frame row 0 = normal
// from geom1 to geom2, dist < 0 = penetration, pos = midpoint (as read back by the env through
// data.contact[i].geom1/geom2, furniture/env/furniture.py:500-513, 1298-1322).
#pragma once
#include "fsim_physics.hpp"

// contact slot (FSIM_CONW: 18 words + 1 pad).  Only the normal is stored; the tangents are rebuilt where needed (fs_frame);
// J*a and J*p of the Newton solve live in registers (SolSlot).
// C_DIST / C_INCM are consumed by fs_make_constraints before it writes C_AREF + 1 / + 2 over them.
enum { C_ACTIVE = 0, C_POS = 1, C_FRAME = 4, C_MU = 7, C_DIM = 8, C_B1 = 9, C_B2 = 10, C_G1 = 11, C_G2 = 12, C_AREF = 13,
       C_DIST = 14, C_INCM = 15, C_DN = 16, C_DT = 17 };

__constant__ int FS_PAIR_MAXCON[12] = {1, 4, 4, 1, 1, 1, 8, 1, 1, 2, 1, 4};

template <class Ctx> struct Emit {
  const Ctx &c;
  int maxn, cg1, cg2;
  float margin, gap;
  __device__ Emit(const Ctx &c_, int mx, int g1, int g2, float mg, float gp) : c(c_), maxn(mx), cg1(g1), cg2(g2), margin(mg), gap(gp) {}
  // Raw contact: position, (unnormalised) normal, distance and the geom pair.  The frame and the pair's solver
  // parameters are filled in afterwards by fs_finish_contacts with one lane per slot (uniform control flow), so the
  // many inlined copies of this call stay small.
  DEV void operator()(int k, float dist, V3 pos, V3 n) const {
    if (k >= maxn) return;
    // a degenerate narrow-phase result (fp32 portal refinement on near-parallel faces) must never reach the solver
    if (!(isfinite(dist) && isfinite(pos.x + pos.y + pos.z) && isfinite(n.x + n.y + n.z)) || dot(n, n) < 1e-12f) return;
    // exact slot allocation: one LDS atomic per emitted contact.  A single wave executes this code, so
    // the allocation order is a deterministic function of the inputs (not of timing).
    int *scal_ = c.I(c.ly.scal);
    int slot = atomicAdd(&scal_[SC_NSLOT], 1);
    if (slot >= c.ly.ncon_max) { scal_[SC_OVERFLOW] |= 2; return; }
    float *r = c.L + c.ly.con + FSIM_CONW * slot;
    int *ri = reinterpret_cast<int *>(r);
    stv3(r + C_POS, pos);
    stv3(r + C_FRAME, n);
    r[C_DIST] = dist;
    r[C_INCM] = margin - gap;
    ri[C_G1] = cg1; ri[C_G2] = cg2;
    ri[C_ACTIVE] = 1;
  }
  // Batched form for the closed-form primitives that know their contact count up front: ONE LDS atomic per pair instead of one
  // per contact (a returning LDS atomic is ~100+ cycles of latency on the single wave's critical path), and the contacts of a
  // pair end up contiguous, in the order MuJoCo lists them.  alloc() returns the first slot (or -1 if none fits) and clips n.
  DEV int alloc(int &n) const {
    int *scal_ = c.I(c.ly.scal);
    int base = atomicAdd(&scal_[SC_NSLOT], n);
    if (base + n > c.ly.ncon_max) { scal_[SC_OVERFLOW] |= 2; n = max(0, c.ly.ncon_max - base); }
    return n > 0 ? base : -1;
  }
  DEV void write(int slot, float dist, V3 pos, V3 n) const {
    float *r = c.L + c.ly.con + FSIM_CONW * slot;
    int *ri = reinterpret_cast<int *>(r);
    const bool ok = isfinite(dist) && isfinite(pos.x + pos.y + pos.z) && isfinite(n.x + n.y + n.z) && dot(n, n) >= 1e-12f;
    stv3(r + C_POS, pos);
    stv3(r + C_FRAME, ok ? n : v3(0, 0, 1));
    r[C_DIST] = ok ? dist : 1.0f;
    r[C_INCM] = margin - gap;
    ri[C_G1] = cg1; ri[C_G2] = cg2;
    ri[C_ACTIVE] = ok ? 1 : 0; // a degenerate result keeps its slot but never reaches the solver
  }
};

// contact frame from the stored unit normal (mju_makeFrame convention: x = n, y = the world y (or z) axis made orthogonal
// to x, z = x cross y)
DEV void fs_frame(const float *r, V3 &x, V3 &y, V3 &z) {
  x = ldv3(r + C_FRAME);
  V3 e = (x.y > -0.5f && x.y < 0.5f) ? v3(0, 1, 0) : v3(0, 0, 1);
  y = normalized(e - x * dot(x, e));
  z = cross(x, y);
}

// lane = slot: unit normal and the geom pair's parameters
template <class Ctx> DEV void fs_finish_contacts(const Ctx &c) {
  CModel &m = c.m;
  int nslot = min(c.I(c.ly.scal)[SC_NSLOT], c.ly.ncon_max);
  for (int s = c.lane; s < nslot; s += 64) {
    float *r = c.L + c.ly.con + FSIM_CONW * s;
    int *ri = reinterpret_cast<int *>(r);
    int cg1 = ri[C_G1], cg2 = ri[C_G2];
    stv3(r + C_FRAME, normalized(ldv3(r + C_FRAME)));
    float mu = fmaxf(GP(m.cg_friction)[3 * cg1], GP(m.cg_friction)[3 * cg2]);
    r[C_MU] = mu;
    int dim = max(m.cg_condim[cg1], m.cg_condim[cg2]);
    if (mu < 1e-15f) dim = 1;
    ri[C_DIM] = dim;
    // body | tree << 8: the tree id rides along so that the solver loops reach the tree's CoM without a dependent lookup
    int b1_ = m.cg_body[cg1], b2_ = m.cg_body[cg2];
    ri[C_B1] = b1_ | (KI(r_tree, b1_) << 8); ri[C_B2] = b2_ | (KI(r_tree, b2_) << 8);
  }
}

// ---- narrow phase primitives ---------------------------------------------------------------
template <class Emit> DEV void np_plane_sphere(const Emit &e, V3 pp, const M3 &pR, V3 sp, float r) {
  V3 n = colv(pR, 2);
  float dist = dot(sp - pp, n) - r;
  if (dist > e.margin) return;
  e(0, dist, sp - n * (r + 0.5f * dist), n);
}
template <class Emit> DEV void np_plane_box(const Emit &e, V3 pp, const M3 &pR, V3 bp, const M3 &bR, V3 size) {
  // corner i = bp +- c0 +- c1 +- c2 (scaled box axes); its height over the plane is dist0 +- a +- b +- cc, so the
  // eight depth tests cost three adds each and the corner itself is only formed for the (<= 4) contacts that are kept
  V3 n = colv(pR, 2);
  float dist0 = dot(bp - pp, n);
  V3 c0 = colv(bR, 0) * size.x, c1 = colv(bR, 1) * size.y, c2 = colv(bR, 2) * size.z;
  float a = dot(n, c0), b = dot(n, c1), cc = dot(n, c2);
  if (dist0 - fabsf(a) - fabsf(b) - fabsf(cc) > e.margin) return; // deepest corner still above the margin
#ifdef FSIM_EMIT_PER_CONTACT
  int cnt = 0;
#pragma unroll 1
  for (int i = 0; i < 8; i++) {
    float sx = (i & 1) ? 1.0f : -1.0f, sy = (i & 2) ? 1.0f : -1.0f, sz_ = (i & 4) ? 1.0f : -1.0f;
    float ld = sx * a + sy * b + sz_ * cc;
    if (dist0 + ld > e.margin || ld > 0) continue;
    float d = dist0 + ld;
    V3 cv = c0 * sx + c1 * sy + c2 * sz_;
    e(cnt, d, cv - n * (0.5f * d) + bp, n);
    if (++cnt >= 4) return;
  }
#else
  // pass 1: which corners are kept (the first four, in corner order, that are below the margin and below the centre)
  int keep = 0, cnt = 0;
#pragma unroll
  for (int i = 0; i < 8; i++) {
    float ld = ((i & 1) ? a : -a) + ((i & 2) ? b : -b) + ((i & 4) ? cc : -cc);
    bool ok = !(dist0 + ld > e.margin || ld > 0) && cnt < 4;
    keep |= ok ? (1 << i) : 0;
    cnt += ok ? 1 : 0;
  }
  if (!cnt) return;
  int nw = cnt;
  const int base = e.alloc(nw);
  if (base < 0) return;
  // pass 2: one slot per kept corner
  int k = 0;
#pragma unroll 1
  for (int mm = keep; mm && k < nw; mm &= mm - 1, k++) {
    int i = __ffs(mm) - 1;
    float sx = (i & 1) ? 1.0f : -1.0f, sy = (i & 2) ? 1.0f : -1.0f, sz_ = (i & 4) ? 1.0f : -1.0f;
    float d = dist0 + sx * a + sy * b + sz_ * cc;
    V3 cv = c0 * sx + c1 * sy + c2 * sz_;
    e.write(base + k, d, cv - n * (0.5f * d) + bp, n);
  }
#endif
}
template <class Emit> DEV void np_plane_cylinder(const Emit &e, V3 pp, const M3 &pR, V3 cp, const M3 &cR, V3 size) {
  V3 n = colv(pR, 2), axis = colv(cR, 2);
  float prjaxis = dot(n, axis);
  if (prjaxis > 0) { axis = -axis; prjaxis = -prjaxis; }
  float dist0 = dot(cp - pp, n);
  // vec = axis (n.axis) - n = minus the part of n perpendicular to the axis.  Written through the cylinder's own x/y axes
  // it has no cancellation: for a disc lying flat, |vec| ~ tilt, and in fp32 the difference of two unit vectors leaves the
  // direction of the lowest rim point (hence the 3-point support triangle) to rounding noise -- the part then never comes
  // to rest (seat of chair_agne_0007 wobbled at 6e-4 rad where the fp64 oracle settles to 1e-7).
  V3 c0 = colv(cR, 0), c1 = colv(cR, 1);
  float a0 = dot(n, c0), a1 = dot(n, c1);
  V3 vec = -(c0 * a0 + c1 * a1);
  float len2 = a0 * a0 + a1 * a1;
  if (len2 >= 1e-12f) vec = vec * (size.x / sqrtf(len2));
  else vec = colv(cR, 0) * size.x;
  float prjvec = dot(vec, n);
  axis = axis * size.y;
  prjaxis *= size.y;
  if (dist0 + prjaxis + prjvec > e.margin) return;
  int cnt = 0;
  float d = dist0 + prjaxis + prjvec;
  e(cnt++, d, cp + vec + axis - n * (0.5f * d), n);
  if (dist0 - prjaxis + prjvec <= e.margin) {
    d = dist0 - prjaxis + prjvec;
    e(cnt++, d, cp + vec - axis - n * (0.5f * d), n);
  }
  float prjvec1 = -0.5f * prjvec;
  if (dist0 + prjaxis + prjvec1 <= e.margin) {
    V3 v1 = normalized(cross(vec, axis)) * (size.x * 0.8660254038f);
    d = dist0 + prjaxis + prjvec1;
    e(cnt++, d, cp + v1 + axis - vec * 0.5f - n * (0.5f * d), n);
    e(cnt++, d, cp - v1 + axis - vec * 0.5f - n * (0.5f * d), n);
  }
}
template <class Emit> DEV void np_sphere_sphere(const Emit &e, V3 p1, float r1, V3 p2, float r2) {
  V3 d = p2 - p1;
  float len = norm(d), dist = len - r1 - r2;
  if (dist > e.margin) return;
  V3 n = len < 1e-15f ? v3(1, 0, 0) : d * (1.0f / len);
  e(0, dist, p1 + n * (r1 + 0.5f * dist), n);
}
template <class Emit> DEV void np_sphere_local(const Emit &e, V3 sp, float r, const M3 &oR, V3 cl, V3 q, bool inside, V3 od, float pen) {
  V3 nl;
  float dist;
  if (!inside) { float d; nl = normalized(q - cl, &d); dist = d - r; }
  else { nl = -od; dist = -pen - r; }
  if (dist > e.margin) return;
  V3 n = mulv(oR, nl);
  e(0, dist, sp + n * (r + 0.5f * dist), n);
}
template <class Emit> DEV void np_sphere_box(const Emit &e, V3 sp, float r, V3 bp, const M3 &bR, V3 size) {
  V3 cl = multv(bR, sp - bp);
  V3 q = v3(fminf(fmaxf(cl.x, -size.x), size.x), fminf(fmaxf(cl.y, -size.y), size.y), fminf(fmaxf(cl.z, -size.z), size.z));
  bool inside = (q.x == cl.x) && (q.y == cl.y) && (q.z == cl.z);
  V3 od = v3(0, 0, 0);
  float pen = 0;
  if (inside) {
    float px = size.x - fabsf(cl.x), py = size.y - fabsf(cl.y), pz = size.z - fabsf(cl.z);
    pen = px; int kb = 0;
    if (py < pen) { pen = py; kb = 1; }
    if (pz < pen) { pen = pz; kb = 2; }
    float sg = comp(cl, kb) >= 0 ? 1.0f : -1.0f;
    od = v3(kb == 0 ? sg : 0, kb == 1 ? sg : 0, kb == 2 ? sg : 0);
  }
  np_sphere_local(e, sp, r, bR, cl, q, inside, od, pen);
}
template <class Emit> DEV void np_sphere_cylinder(const Emit &e, V3 sp, float r, V3 cp, const M3 &cR, V3 size) {
  V3 cl = multv(cR, sp - cp);
  float rho = sqrtf(cl.x * cl.x + cl.y * cl.y), rc = size.x, h = size.y;
  bool inside = (rho <= rc) && (fabsf(cl.z) <= h);
  V3 od = v3(0, 0, 0), q = cl;
  float pen = 0;
  if (inside) {
    float pr = rc - rho, pz = h - fabsf(cl.z);
    if (pz < pr) { pen = pz; od.z = cl.z >= 0 ? 1.0f : -1.0f; }
    else { pen = pr; if (rho > 1e-15f) { od.x = cl.x / rho; od.y = cl.y / rho; } else od.x = 1; }
  } else {
    float sc = rho > rc ? rc / rho : 1.0f;
    q = v3(cl.x * sc, cl.y * sc, fminf(fmaxf(cl.z, -h), h));
  }
  np_sphere_local(e, sp, r, cR, cl, q, inside, od, pen);
}

// dynamic pick of one of three values without a (scratch-backed) indexed local array
DEV V3 pick3(V3 a, V3 b, V3 c, int k) { return k == 0 ? a : (k == 1 ? b : c); }
DEV float sz(V3 s, int k) { return comp(s, k); }

// box-box: 15-axis separating-axis test, then either one edge-edge contact or the face manifold
//   reference face rectangle  (x) incident face quad, both projected on the reference face:
//   manifold = {quad vertices inside the rectangle} + {quad edge x rectangle side crossings} + {rectangle corners
//   strictly inside the quad}  (the vertex set of the clipped polygon, <= 8 points, no polygon buffers needed).
template <class Emit> DEV void np_box_box(const Emit &e, V3 p1, const M3 &R1, V3 s1, V3 p2, const M3 &R2, V3 s2) {
  const V3 A0 = colv(R1, 0), A1 = colv(R1, 1), A2 = colv(R1, 2), B0 = colv(R2, 0), B1 = colv(R2, 1), B2 = colv(R2, 2);
  const V3 d = p2 - p1;
  const float margin = e.margin;
  float best = -1e30f; int bestType = 0, bk = 0; float bsgn = 1;
  float bestE = -1e30f; int ei = 0, ej = 0; V3 eL = v3(0, 0, 0);
#pragma unroll
  for (int i = 0; i < 3; i++) {
    V3 Ai = i == 0 ? A0 : (i == 1 ? A1 : A2);
    float c0 = fabsf(dot(Ai, B0)) + 1e-7f, c1 = fabsf(dot(Ai, B1)) + 1e-7f, c2 = fabsf(dot(Ai, B2)) + 1e-7f;
    float t = dot(d, Ai);
    float sep = fabsf(t) - (sz(s1, i) + s2.x * c0 + s2.y * c1 + s2.z * c2);
    if (sep > margin) return;
    if (sep > best) { best = sep; bestType = 0; bk = i; bsgn = t < 0 ? -1.0f : 1.0f; }
  }
#pragma unroll
  for (int j = 0; j < 3; j++) {
    V3 Bj = j == 0 ? B0 : (j == 1 ? B1 : B2);
    float c0 = fabsf(dot(A0, Bj)) + 1e-7f, c1 = fabsf(dot(A1, Bj)) + 1e-7f, c2 = fabsf(dot(A2, Bj)) + 1e-7f;
    float t = dot(d, Bj);
    float sep = fabsf(t) - (sz(s2, j) + s1.x * c0 + s1.y * c1 + s1.z * c2);
    if (sep > margin) return;
    if (sep > best) { best = sep; bestType = 1; bk = j; bsgn = t < 0 ? -1.0f : 1.0f; }
  }
#pragma unroll 1
  for (int ij = 0; ij < 9; ij++) {
    int i = ij / 3, j = ij - 3 * i;
    V3 Lx = cross(pick3(A0, A1, A2, i), pick3(B0, B1, B2, j));
    float len = norm(Lx);
    if (len < 1e-6f) continue;
    Lx = Lx * (1.0f / len);
    float t = dot(d, Lx);
    float ra = s1.x * fabsf(dot(A0, Lx)) + s1.y * fabsf(dot(A1, Lx)) + s1.z * fabsf(dot(A2, Lx));
    float rb = s2.x * fabsf(dot(B0, Lx)) + s2.y * fabsf(dot(B1, Lx)) + s2.z * fabsf(dot(B2, Lx));
    float sep = fabsf(t) - (ra + rb);
    if (sep > margin) return;
    if (sep > bestE) { bestE = sep; ei = i; ej = j; eL = Lx * (t < 0 ? -1.0f : 1.0f); }
  }
  if (bestE > best + 1e-6f + 0.05f * fabsf(best)) {
    // edge-edge contact: closest points of the two support edges along eL
    V3 n = eL, pa = p1, pb = p2;
#pragma unroll
    for (int k = 0; k < 3; k++) {
      V3 Ak = k == 0 ? A0 : (k == 1 ? A1 : A2), Bk = k == 0 ? B0 : (k == 1 ? B1 : B2);
      if (k != ei) pa = pa + Ak * (dot(Ak, n) > 0 ? sz(s1, k) : -sz(s1, k));
      if (k != ej) pb = pb + Bk * (dot(Bk, n) > 0 ? -sz(s2, k) : sz(s2, k));
    }
    V3 Ae = pick3(A0, A1, A2, ei), Be = pick3(B0, B1, B2, ej);
    float sae = sz(s1, ei), sbe = sz(s2, ej);
    V3 w = pa - pb;
    float b = dot(Ae, Be), dd = dot(Ae, w), ee = dot(Be, w), den = 1 - b * b;
    float sa = 0, tb = 0;
    if (den > 1e-12f) { sa = (b * ee - dd) / den; tb = (ee - b * dd) / den; }
    sa = fminf(fmaxf(sa, -sae), sae);
    tb = fminf(fmaxf(tb, -sbe), sbe);
    pa = pa + Ae * sa; pb = pb + Be * tb;
    e(0, bestE, (pa + pb) * 0.5f, n);
    return;
  }
  // face contact.  reference box r (face axis ka, outward normal nr towards the incident box), incident box i
  const bool f0 = bestType == 0;
  const V3 pr = f0 ? p1 : p2, pi_ = f0 ? p2 : p1, sr = f0 ? s1 : s2, si = f0 ? s2 : s1;
  const V3 Rr0 = f0 ? A0 : B0, Rr1 = f0 ? A1 : B1, Rr2 = f0 ? A2 : B2, Ri0 = f0 ? B0 : A0, Ri1 = f0 ? B1 : A1, Ri2 = f0 ? B2 : A2;
  const int ka = bk;
  const V3 nr = pick3(Rr0, Rr1, Rr2, ka) * (f0 ? bsgn : -bsgn);
  int kinc = 0; float mind = 1e30f, sgninc = 1;
#pragma unroll
  for (int k = 0; k < 3; k++) {
    float t = dot(k == 0 ? Ri0 : (k == 1 ? Ri1 : Ri2), nr);
    if (-fabsf(t) < mind) { mind = -fabsf(t); kinc = k; sgninc = t > 0 ? -1.0f : 1.0f; }
  }
  const int k1 = (kinc + 1) % 3, k2 = (kinc + 2) % 3, u1 = (ka + 1) % 3, u2 = (ka + 2) % 3;
  const V3 Rik = pick3(Ri0, Ri1, Ri2, kinc), E1 = pick3(Ri0, Ri1, Ri2, k1) * sz(si, k1), E2 = pick3(Ri0, Ri1, Ri2, k2) * sz(si, k2);
  const V3 U1 = pick3(Rr0, Rr1, Rr2, u1), U2 = pick3(Rr0, Rr1, Rr2, u2);
  const float hu = sz(sr, u1), hv = sz(sr, u2);
  const V3 fc = pi_ + Rik * (sgninc * sz(si, kinc));
  const V3 rc = pr + nr * sz(sr, ka);
  // incident quad in reference-face coordinates: centre (cx, cy), edge vectors (e1x, e1y), (e2x, e2y);
  // vertex q = centre + sx*e1 + sy*e2 with (sx, sy) = (1,1), (-1,1), (-1,-1), (1,-1)
  const V3 rcen = fc - rc;
  const float cx = dot(rcen, U1), cy = dot(rcen, U2), e1x = dot(E1, U1), e1y = dot(E1, U2), e2x = dot(E2, U1), e2y = dot(E2, U2);
  const V3 ninc = Rik * sgninc;
  const float denom = dot(ninc, nr);
  const V3 nout = f0 ? nr : -nr;
  int cnt = 0;
  auto lift = [&](float x, float y) {
    V3 q = rc + U1 * x + U2 * y;
    float h = fabsf(denom) > 1e-9f ? dot(fc - q, ninc) / denom : 0.0f;
    if (h > margin) return;
    e(cnt, h, q + nr * (0.5f * h), nout);
    cnt++;
  };
#pragma unroll 1
  for (int q = 0; q < 4; q++) {
    float sx = (q == 0 || q == 3) ? 1.0f : -1.0f, sy = q < 2 ? 1.0f : -1.0f;
    float ax = cx + sx * e1x + sy * e2x, ay = cy + sx * e1y + sy * e2y;
    if (fabsf(ax) <= hu && fabsf(ay) <= hv) lift(ax, ay);
    // edge q -> q+1 against the four rectangle sides
    int q1 = (q + 1) & 3;
    float tx = (q1 == 0 || q1 == 3) ? 1.0f : -1.0f, ty = q1 < 2 ? 1.0f : -1.0f;
    float bx = cx + tx * e1x + ty * e2x, by = cy + tx * e1y + ty * e2y;
#pragma unroll 1
    for (int side = 0; side < 4; side++) {
      bool xs = side < 2;                       // sides 0,1: x = +-hu;  sides 2,3: y = +-hv
      float sg = (side & 1) ? -1.0f : 1.0f, lim = xs ? hu : hv, olim = xs ? hv : hu;
      float da = sg * (xs ? ax : ay) - lim, db = sg * (xs ? bx : by) - lim;
      if (!((da < 0 && db > 0) || (da > 0 && db < 0))) continue;
      float t = da / (da - db);
      float px = ax + t * (bx - ax), py = ay + t * (by - ay);
      float o = fabsf(xs ? py : px);
      if (xs ? (o <= olim) : (o < olim)) lift(px, py); // corner crossings belong to the x sides only
    }
  }
  // rectangle corners strictly inside the quad: |local coords| < 1 in the quad's own (e1, e2) basis
  float det = e1x * e2y - e1y * e2x;
  if (fabsf(det) > 1e-12f) {
    float idet = 1.0f / det;
#pragma unroll 1
    for (int q = 0; q < 4; q++) {
      float x = ((q == 0 || q == 3) ? hu : -hu) - cx, y = (q < 2 ? hv : -hv) - cy;
      float a = (x * e2y - y * e2x) * idet, b = (e1x * y - e1y * x) * idet;
      if (fabsf(a) < 1.0f && fabsf(b) < 1.0f) lift(x + cx, y + cy);
    }
  }
}

// distance between the axis segments p1 +- h1*a1 and p2 +- h2*a2 (closest points of two segments, clamped)
DEV float np_capsule_gap(V3 p1, V3 a1, float h1, V3 p2, V3 a2, float h2) {
  V3 d1 = a1 * (2 * h1), d2 = a2 * (2 * h2), q1 = p1 - a1 * h1, q2 = p2 - a2 * h2, r = q1 - q2;
  float A = dot(d1, d1), E = dot(d2, d2), F = dot(d2, r), C = dot(d1, r), B = dot(d1, d2), den = A * E - B * B;
  float sN = den > 1e-12f ? fminf(fmaxf((B * F - C * E) / den, 0.0f), 1.0f) : 0.0f;
  float tN = E > 1e-12f ? (B * sN + F) / E : 0.0f;
  if (tN < 0) { tN = 0; sN = A > 1e-12f ? fminf(fmaxf(-C / A, 0.0f), 1.0f) : 0.0f; }
  else if (tN > 1) { tN = 1; sN = A > 1e-12f ? fminf(fmaxf((B - C) / A, 0.0f), 1.0f) : 0.0f; }
  V3 dv = r + d1 * sN - d2 * tN;
  return norm(dv);
}

// conservative separating-axis pre-test for two cylinders: both axes, their cross product, the two radial directions
// towards the other centre and the centre line (e.g. a link cylinder resting 7 mm above the flat top of the base)
DEV bool np_cyl_cyl_separated(V3 p1, V3 a1, V3 s1, V3 p2, V3 a2, V3 s2, float margin) {
  const V3 d = p2 - p1;
  bool sep = false;
#pragma unroll
  for (int k = 0; k < 6; k++) {
    V3 Lx = k == 0 ? a1 : (k == 1 ? a2 : (k == 2 ? cross(a1, a2) : (k == 3 ? d - a1 * dot(d, a1) : (k == 4 ? d - a2 * dot(d, a2) : d))));
    float ln = norm(Lx);
    if (ln < 1e-9f) continue;
    Lx = Lx * (1.0f / ln);
    float c1 = dot(a1, Lx), c2 = dot(a2, Lx);
    float r1 = s1.y * fabsf(c1) + s1.x * sqrtf(fmaxf(1.0f - c1 * c1, 0.0f)), r2 = s2.y * fabsf(c2) + s2.x * sqrtf(fmaxf(1.0f - c2 * c2, 0.0f));
    if (fabsf(dot(d, Lx)) > r1 + r2 + margin) sep = true;
  }
  return sep;
}

// conservative separating-axis pre-test for cylinder (geom 1) vs box (geom 2): the box face normals, the cylinder axis and
// the radial direction towards the box centre.  true => some axis separates the two by more than the margin, no contact
DEV bool np_cyl_box_separated(V3 pc, const M3 &Rc, V3 sc, V3 pb, const M3 &Rb, V3 sb, float margin) {
  const V3 a = colv(Rc, 2), d = pb - pc, B0 = colv(Rb, 0), B1 = colv(Rb, 1), B2 = colv(Rb, 2);
  V3 rad = d - a * dot(d, a);
  float rl = norm(rad);
  rad = rl > 1e-9f ? rad * (1.0f / rl) : B0;
  bool sep = false;
#pragma unroll
  for (int k = 0; k < 5; k++) {
    V3 Lx = k == 0 ? B0 : (k == 1 ? B1 : (k == 2 ? B2 : (k == 3 ? a : rad)));
    float al = dot(a, Lx);
    float rc = sc.y * fabsf(al) + sc.x * sqrtf(fmaxf(1.0f - al * al, 0.0f));
    float rb = sb.x * fabsf(dot(B0, Lx)) + sb.y * fabsf(dot(B1, Lx)) + sb.z * fabsf(dot(B2, Lx));
    if (fabsf(dot(d, Lx)) > rc + rb + margin) sep = true;
  }
  return sep;
}

// the hull vertex furthest along the local direction dl (arg max; first one on ties: a function of the inputs alone)
typedef const __attribute__((address_space(1))) float *MeshVerts; // (global memory: the model's hull-vertex table)
DEV V3 np_mesh_support(MeshVerts verts, int n, V3 dl) {
  float best = -3.0e38f;
  V3 bv = v3(0, 0, 0);
  for (int i = 0; i < n; i++) {
    const V3 v = v3(verts[3 * i], verts[3 * i + 1], verts[3 * i + 2]);
    const float d = dot(v, dl);
    if (d > best) { best = d; bv = v; }
  }
  return bv;
}
// plane - convex mesh: the hull's lowest vertex, then the vertices a slightly tilted "down" picks in three directions 120 degrees
// apart (for a hull resting on a face these are spread over that face: a stable support polygon instead of three neighbours of the
// lowest corner) -- the scheme of MuJoCo's mjc_PlaneConvex; every distinct vertex within the margin is a contact, at most four.
template <class Emit> DEV void np_plane_mesh(const Emit &e, V3 pp, const M3 &pR, V3 mp, const M3 &mR, MeshVerts verts, int n) {
  const V3 nw = colv(pR, 2), t1 = colv(pR, 0), t2 = colv(pR, 1);
  V3 prev[4];
  int cnt = 0;
  for (int k = 0; k < 4; k++) {
    V3 dw = -nw;
    if (k > 0) { const float a = 2.0943951f * (float)(k - 1); dw = dw + (t1 * cosf(a) + t2 * sinf(a)) * 1e-3f; }
    const V3 vl = np_mesh_support(verts, n, multv(mR, dw));
    bool dup = false;
    for (int j = 0; j < cnt; j++) if (prev[j].x == vl.x && prev[j].y == vl.y && prev[j].z == vl.z) dup = true;
    if (dup) continue;
    prev[cnt++] = vl;
    const V3 vw = mulv(mR, vl) + mp;
    const float dist = dot(vw - pp, nw);
    if (dist > e.margin) continue;
    e(k, dist, vw - nw * (0.5f * dist), nw);
  }
}

// ---- Minkowski portal refinement for cylinder-box / cylinder-cylinder; every pair with a capsule or a convex mesh in it ---------
struct Shape { int type; V3 pos; M3 R; V3 size; MeshVerts verts; int nvert; }; // verts / nvert: hull of a convex mesh (geom frame)
// MESH: the kernel serves models with convex-mesh colliders (the generic kernels; the kernels specialised for one model -- the
// benchmark's -- are compiled without that branch: nothing that is not the benchmark's work belongs in its substep loop)
template <bool MESH> DEV V3 np_support(const Shape &s, V3 dir) {
  V3 dl = multv(s.R, dir), pl;
  if (MESH && s.type == GT_MESH) pl = np_mesh_support(s.verts, s.nvert, dl);
  else if (s.type == GT_BOX) pl = v3(dl.x >= 0 ? s.size.x : -s.size.x, dl.y >= 0 ? s.size.y : -s.size.y, dl.z >= 0 ? s.size.z : -s.size.z);
  else if (s.type == GT_CYLINDER) {
    float rho = sqrtf(dl.x * dl.x + dl.y * dl.y);
    pl = rho > 1e-12f ? v3(dl.x / rho * s.size.x, dl.y / rho * s.size.x, 0) : v3(0, 0, 0);
    pl.z = dl.z >= 0 ? s.size.y : -s.size.y;
  } else { // sphere; capsule = sphere swept along the local z axis
    float n = norm(dl);
    pl = n > 1e-12f ? dl * (s.size.x / n) : v3(0, 0, 0);
    if (s.type == GT_CAPSULE) pl.z += dl.z >= 0 ? s.size.y : -s.size.y;
  }
  return mulv(s.R, pl) + s.pos;
}
struct Sup { V3 v, a, b; };
template <bool MESH> DEV Sup np_msup_t(const Shape &A, const Shape &B, V3 dir) { Sup s; s.a = np_support<MESH>(A, -dir); s.b = np_support<MESH>(B, dir); s.v = s.b - s.a; return s; }
template <class Emit, bool MESH = false> DEV void np_mpr(const Emit &e, const Shape &A, const Shape &B) {
  auto np_msup = [](const Shape &A_, const Shape &B_, V3 d_) { return np_msup_t<MESH>(A_, B_, d_); };
  Sup v0, v1, v2, v3_, v4;
  v0.a = A.pos; v0.b = B.pos; v0.v = v0.b - v0.a;
  if (dot(v0.v, v0.v) < 1e-20f) v0.v.x = 1e-5f;
  V3 n = normalized(-v0.v);
  v1 = np_msup(A, B, n);
  if (dot(v1.v, n) <= 0) return;
  n = cross(v1.v, v0.v);
  if (dot(n, n) < 1e-20f) {
    n = normalized(v1.v - v0.v);
    e(0, -dot(v1.v, n), (v1.a + v1.b) * 0.5f, -n);
    return;
  }
  n = normalized(n);
  v2 = np_msup(A, B, n);
  if (dot(v2.v, n) <= 0) return;
  n = cross(v1.v - v0.v, v2.v - v0.v);
  if (dot(n, v0.v) > 0) { Sup t = v1; v1 = v2; v2 = t; n = -n; }
  for (int it = 0;; it++) {
    if (it > 64) return;
    n = normalized(n);
    v3_ = np_msup(A, B, n);
    if (dot(v3_.v, n) <= 0) return;
    if (dot(cross(v1.v, v3_.v), v0.v) < 0) { v2 = v3_; n = cross(v1.v - v0.v, v3_.v - v0.v); continue; }
    if (dot(cross(v3_.v, v2.v), v0.v) < 0) { v1 = v3_; n = cross(v3_.v - v0.v, v2.v - v0.v); continue; }
    break;
  }
  bool hit = false;
  for (int it = 0; it < 64; it++) {
    n = cross(v2.v - v1.v, v3_.v - v1.v);
    float ln;
    n = normalized(n, &ln);
    if (ln < 1e-30f) return;
    float dpl = dot(n, v1.v);
    if (dpl >= 0) hit = true;
    v4 = np_msup(A, B, n);
    float delta = dot(v4.v, n) - dot(v3_.v, n);
    if (dot(v4.v, n) < 0 && !hit) return;
    if (delta <= 1e-6f || it == 63) {
      if (!hit) return;
      float b0 = dot(cross(v1.v, v2.v), v3_.v), b1 = dot(cross(v3_.v, v2.v), v0.v), b2 = dot(cross(v0.v, v1.v), v3_.v), b3 = dot(cross(v2.v, v1.v), v0.v);
      float sum = b0 + b1 + b2 + b3;
      if (sum <= 0) {
        b0 = 0; b1 = dot(cross(v2.v, v3_.v), n); b2 = dot(cross(v3_.v, v1.v), n); b3 = dot(cross(v1.v, v2.v), n);
        sum = b1 + b2 + b3;
      }
      V3 pa, pb;
      if (sum > 1e-30f) {
        float inv = 1.0f / sum;
        pa = (v0.a * b0 + v1.a * b1 + v2.a * b2 + v3_.a * b3) * inv;
        pb = (v0.b * b0 + v1.b * b1 + v2.b * b2 + v3_.b * b3) * inv;
      } else { // degenerate portal: fall back to the centroid of the portal's support points
        pa = (v1.a + v2.a + v3_.a) * (1.0f / 3.0f);
        pb = (v1.b + v2.b + v3_.b) * (1.0f / 3.0f);
      }
      e(0, -dpl, (pa + pb) * 0.5f, -n);
      return;
    }
    V3 cr = cross(v4.v, v0.v);
    if (dot(v1.v, cr) > 0) { if (dot(v2.v, cr) > 0) v1 = v4; else v3_ = v4; }
    else { if (dot(v3_.v, cr) > 0) v2 = v4; else v1 = v4; }
  }
}

// ---- driver ----------------------------------------------------------------------------------
#ifdef FSIM_PROFILE
#define FS_CPROF(slot) do { long long t1c_ = clock64(); if (c.lane == 0) c.I(c.ly.scal)[16 + slot] += (int)((t1c_ - t0c_) >> 4); t0c_ = t1c_; } while (0)
#else
#define FS_CPROF(slot) do { } while (0)
#endif
template <class Ctx> DEV void fs_collide(const Ctx &c) {
  CModel &m = c.m;
  float *L = c.L;
  int *scal = c.I(c.ly.scal);
#ifdef FSIM_PROFILE
  long long t0c_ = clock64();
#endif
  for (int g = c.lane; g < c.D.ncg; g += 64) {
    int b = m.cg_body[g];
    M3 Rb = ldm3(L + c.ly.xmat + 9 * b);
    V3 gp = ldv3(L + c.ly.xpos + 3 * b) + mulv(Rb, ldv3(GP(m.cg_pos) + 3 * g));
    if (c.D.agent == 2) { // cursor boxes are world geoms whose body_pos the env rewrites (furniture.py:3139)
      int cm = m.cg_cursor[g];
      if (cm) { int k = (cm & 1) ? 0 : 1; gp = gp + ldv3(L + c.ly.env + E_GROUP + c.D.nparts + EC_POS + 3 * k) - ldv3(GP(m.cursor_pos0) + 3 * k); }
    }
    stv3(L + c.ly.gpos + 3 * g, gp);
    stm3(L + c.ly.gmat + 9 * g, mulm(Rb, ldm3(GP(m.cg_mat) + 9 * g)));
  }
  SYNC();
  FS_CPROF(29);
  // broadphase + ordered compaction, two stages so that the expensive test only runs on the few pairs that need it (a
  // wave pays for a branch as soon as ONE lane takes it):
  //   stage 1, all candidate pairs, 64 per pass: collision masks + bounding spheres (plane: signed distance) -> list A
  //            (kept in the contact-slot area, which is dead until the narrow phase emits);
  //   stage 2, list A: exact point-to-solid distance for flat / long shapes -> the survivor list of the narrow phase.
  // Both compactions are ballot-ordered, so survivors (hence contacts) stay in candidate-pair order.
  int *surv = c.I(c.ly.surv);
  int *listA = c.I(c.ly.con);
  const int capA = FSIM_CONW * c.ly.ncon_max;
  const int *ctype = c.I(c.ly.contype), *caff = c.I(c.ly.conaff);
  int nA = 0;
  {
    // stage 1 is latency, not arithmetic: per pass ONE 8-byte record per lane (fetched a pass ahead) and ONE batch of LDS reads
    // issued unconditionally (masks, both centres, the plane normal) -- no load waits behind a branch
    typedef int i2_t __attribute__((ext_vector_type(2)));
    const int last = c.D.ncp - 1;
    i2_t rec = GPC<i2_t>(m.pair_bp)[min(c.lane, last)];
    for (int p0 = 0; p0 < c.D.ncp; p0 += 64) {
      const int p = p0 + c.lane;
      const i2_t cur = rec;
      if (p0 + 64 < c.D.ncp) rec = GPC<i2_t>(m.pair_bp)[min(p + 64, last)];
      const int g1 = cur.x & 255, g2 = (cur.x >> 8) & 255;
      const bool plane = (cur.x >> 16) & 1;
      const float bound = __int_as_float(cur.y);
      const int ct1 = ctype[g1], ca1 = caff[g1], ct2 = ctype[g2], ca2 = caff[g2];
      const V3 d = ldv3(L + c.ly.gpos + 3 * g2) - ldv3(L + c.ly.gpos + 3 * g1);
      const V3 n = v3(L[c.ly.gmat + 9 * g1 + 2], L[c.ly.gmat + 9 * g1 + 5], L[c.ly.gmat + 9 * g1 + 8]);
      const bool near = plane ? dot(d, n) <= bound : dot(d, d) <= bound * bound;
      const bool pass = p <= last && ((ct1 & ca2) | (ct2 & ca1)) != 0 && near;
      unsigned long long mask = __ballot(pass);
      int idx = nA + __popcll(mask & ((1ull << c.lane) - 1ull));
      if (pass && idx < capA) listA[idx] = p;
      nA += __popcll(mask);
    }
  }
  if (nA > capA) { nA = capA; if (c.lane == 0) scal[SC_OVERFLOW] |= 1; }
  SYNC();
  int nsurv = 0;
  for (int i0 = 0; i0 < nA; i0 += 64) {
    const int i = i0 + c.lane;
    bool pass = false;
    int p = 0;
    {
      // the whole 64-byte pair record and both geom poses are fetched up front (one global + one LDS round trip for the pass)
      p = listA[min(i, max(nA - 1, 0))];
      const i4_t q0 = GPC<i4_t>(m.pair_rec)[4 * p];
      const f4_t q1 = GPC<f4_t>(m.pair_rec)[4 * p + 1], q2 = GPC<f4_t>(m.pair_rec)[4 * p + 2], q3 = GPC<f4_t>(m.pair_rec)[4 * p + 3];
      const int g1 = q0.x, g2 = q0.y, t1 = q0.w & 255, t2 = q0.w >> 8;
      const float margin = q1.x, r1 = q1.z, r2 = q1.w;
      const V3 d = ldv3(L + c.ly.gpos + 3 * g2) - ldv3(L + c.ly.gpos + 3 * g1);
      const M3 Ra = ldm3(L + c.ly.gmat + 9 * g1), Rb = ldm3(L + c.ly.gmat + 9 * g2);
      pass = i < nA;
      // tighter test for flat / long shapes (a 0.64 x 0.24 x 0.04 table top has a 0.34 m bounding sphere): the distance from
      // the OTHER geom's centre to this box / cylinder (exact point-solid distance) must be within the other geom's
      // bounding radius.  Conservative: never rejects a pair that can touch.
      if (t1 != GT_PLANE) {
#pragma unroll
        for (int side = 0; side < 2; side++) {
          const int ty = side ? t1 : t2;                       // solid tested
          const float ro = (side ? r2 : r1) + margin;         // other geom's radius
          const V3 dw = side ? -d : d;                        // centre(solid) - centre(other)
          const M3 &R = side ? Ra : Rb;
          const V3 cl = v3(-(R.m[0] * dw.x + R.m[3] * dw.y + R.m[6] * dw.z), -(R.m[1] * dw.x + R.m[4] * dw.y + R.m[7] * dw.z), -(R.m[2] * dw.x + R.m[5] * dw.y + R.m[8] * dw.z));
          const f4_t qs = side ? q2 : q3;
          const V3 sz_ = v3(qs.x, qs.y, qs.z);
          const V3 e = v3(fmaxf(fabsf(cl.x) - sz_.x, 0.0f), fmaxf(fabsf(cl.y) - sz_.y, 0.0f), fmaxf(fabsf(cl.z) - sz_.z, 0.0f));
          const float er = fmaxf(sqrtf(cl.x * cl.x + cl.y * cl.y) - sz_.x, 0.0f), ez = fmaxf(fabsf(cl.z) - sz_.y, 0.0f);
          const float dist2 = ty == GT_BOX ? dot(e, e) : er * er + ez * ez;
          if ((ty == GT_BOX || ty == GT_CYLINDER) && dist2 > ro * ro) pass = false;
        }
      }
    }
    unsigned long long mask = __ballot(pass);
    int idx = nsurv + __popcll(mask & ((1ull << c.lane) - 1ull));
    if (pass && idx < c.ly.maxsurv) surv[idx] = p;
    nsurv += __popcll(mask);
  }
  if (nsurv > c.ly.maxsurv) { nsurv = c.ly.maxsurv; if (c.lane == 0) scal[SC_OVERFLOW] |= 1; }
  SYNC();
  if (c.lane == 0) { scal[SC_NSURV] = nsurv; scal[SC_NSLOT] = 0; }
  SYNC();
  FS_CPROF(30);
  for (int i = c.lane; i < nsurv; i += 64) {
    int p = surv[i];
    const i4_t q0 = GPC<i4_t>(m.pair_rec)[4 * p];
    const f4_t q1 = GPC<f4_t>(m.pair_rec)[4 * p + 1], q2 = GPC<f4_t>(m.pair_rec)[4 * p + 2],
                 q3 = GPC<f4_t>(m.pair_rec)[4 * p + 3];
    const int g1 = q0.x, g2 = q0.y, pt = q0.z;
    const float margin = q1.x, gap = q1.y;
    Emit<Ctx> e(c, FS_PAIR_MAXCON[pt], g1, g2, margin, gap);
    V3 p1 = ldv3(L + c.ly.gpos + 3 * g1), p2 = ldv3(L + c.ly.gpos + 3 * g2);
    M3 R1 = ldm3(L + c.ly.gmat + 9 * g1), R2 = ldm3(L + c.ly.gmat + 9 * g2);
    V3 s1 = v3(q2.x, q2.y, q2.z), s2 = v3(q3.x, q3.y, q3.z);
#ifdef FSIM_NPPROF
    // development: run the pair types one after the other so that each type's path length can be timed on its own
    for (int ptq = 0; ptq < 9; ptq++) {
      long long tq0_ = clock64();
      if (pt == ptq) {
    switch (pt) {
      case PT_PLANE_SPHERE: np_plane_sphere(e, p1, R1, p2, s2.x); break;
      case PT_PLANE_BOX: np_plane_box(e, p1, R1, p2, R2, s2); break;
      case PT_PLANE_CYL: np_plane_cylinder(e, p1, R1, p2, R2, s2); break;
      case PT_SPHERE_SPHERE: np_sphere_sphere(e, p1, s1.x, p2, s2.x); break;
      case PT_SPHERE_BOX: np_sphere_box(e, p1, s1.x, p2, R2, s2); break;
      case PT_SPHERE_CYL: np_sphere_cylinder(e, p1, s1.x, p2, R2, s2); break;
      case PT_BOX_BOX: np_box_box(e, p1, R1, s1, p2, R2, s2); break;
      case PT_PLANE_CAP: { // the capsule's two end spheres against the plane (mjc_PlaneCapsule)
        const V3 ax = colv(R2, 2);
        np_plane_sphere(e, p1, R1, p2 + ax * s2.y, s2.x);
        np_plane_sphere(e, p1, R1, p2 - ax * s2.y, s2.x);
        break;
      }
      case PT_PLANE_MESH: { if constexpr (!Ctx::PLAIN) { const int mk = __float_as_int(q3.w); np_plane_mesh(e, p1, R1, p2, R2, GP(m.mesh_vert) + 3 * (mk & 0xffff), mk >> 16); } break; }
      default: {
        // cylinder c capsule of the same radius and half length: if the two capsules are farther apart than the
        // margin the cylinders cannot touch (robot link pairs that sit next to each other but never collide)
        if (pt == PT_CYL_CYL && (np_capsule_gap(p1, colv(R1, 2), s1.y, p2, colv(R2, 2), s2.y) - s1.x - s2.x > margin ||
                                 np_cyl_cyl_separated(p1, colv(R1, 2), s1, p2, colv(R2, 2), s2, margin))) break;
        if (pt == PT_CYL_BOX && np_cyl_box_separated(p1, R1, s1, p2, R2, s2, margin)) break;
        Shape A, B;
        A.type = q0.w & 255; A.pos = p1; A.R = R1; A.size = s1;
        B.type = q0.w >> 8; B.pos = p2; B.R = R2; B.size = s2;
        A.verts = nullptr; A.nvert = 0; B.verts = nullptr; B.nvert = 0;
        if constexpr (!Ctx::PLAIN) { const int ma = __float_as_int(q2.w), mb = __float_as_int(q3.w); // (0 unless the geom is a convex mesh)
          A.verts = GP(m.mesh_vert) + 3 * (ma & 0xffff); A.nvert = ma >> 16; B.verts = GP(m.mesh_vert) + 3 * (mb & 0xffff); B.nvert = mb >> 16; }
        np_mpr<Emit<Ctx>, !Ctx::PLAIN>(e, A, B);
      }
    }
      }
      long long dq_ = clock64() - tq0_;
      int slotq_ = ptq == PT_PLANE_BOX ? 53 : (ptq == PT_BOX_BOX ? 54 : (ptq >= PT_CYL_BOX ? 48 : 49));
      if (c.lane == 0) scal[slotq_] += (int)(dq_ >> 4);
    }
#else
    switch (pt) {
      case PT_PLANE_SPHERE: np_plane_sphere(e, p1, R1, p2, s2.x); break;
      case PT_PLANE_BOX: np_plane_box(e, p1, R1, p2, R2, s2); break;
      case PT_PLANE_CYL: np_plane_cylinder(e, p1, R1, p2, R2, s2); break;
      case PT_SPHERE_SPHERE: np_sphere_sphere(e, p1, s1.x, p2, s2.x); break;
      case PT_SPHERE_BOX: np_sphere_box(e, p1, s1.x, p2, R2, s2); break;
      case PT_SPHERE_CYL: np_sphere_cylinder(e, p1, s1.x, p2, R2, s2); break;
      case PT_BOX_BOX: np_box_box(e, p1, R1, s1, p2, R2, s2); break;
      case PT_PLANE_CAP: { // the capsule's two end spheres against the plane (mjc_PlaneCapsule)
        const V3 ax = colv(R2, 2);
        np_plane_sphere(e, p1, R1, p2 + ax * s2.y, s2.x);
        np_plane_sphere(e, p1, R1, p2 - ax * s2.y, s2.x);
        break;
      }
      case PT_PLANE_MESH: { if constexpr (!Ctx::PLAIN) { const int mk = __float_as_int(q3.w); np_plane_mesh(e, p1, R1, p2, R2, GP(m.mesh_vert) + 3 * (mk & 0xffff), mk >> 16); } break; }
      default: {
        // cylinder c capsule of the same radius and half length: if the two capsules are farther apart than the
        // margin the cylinders cannot touch (robot link pairs that sit next to each other but never collide)
        if (pt == PT_CYL_CYL && (np_capsule_gap(p1, colv(R1, 2), s1.y, p2, colv(R2, 2), s2.y) - s1.x - s2.x > margin ||
                                 np_cyl_cyl_separated(p1, colv(R1, 2), s1, p2, colv(R2, 2), s2, margin))) break;
        if (pt == PT_CYL_BOX && np_cyl_box_separated(p1, R1, s1, p2, R2, s2, margin)) break;
        Shape A, B;
        A.type = q0.w & 255; A.pos = p1; A.R = R1; A.size = s1;
        B.type = q0.w >> 8; B.pos = p2; B.R = R2; B.size = s2;
        A.verts = nullptr; A.nvert = 0; B.verts = nullptr; B.nvert = 0;
        if constexpr (!Ctx::PLAIN) { const int ma = __float_as_int(q2.w), mb = __float_as_int(q3.w); // (0 unless the geom is a convex mesh)
          A.verts = GP(m.mesh_vert) + 3 * (ma & 0xffff); A.nvert = ma >> 16; B.verts = GP(m.mesh_vert) + 3 * (mb & 0xffff); B.nvert = mb >> 16; }
        np_mpr<Emit<Ctx>, !Ctx::PLAIN>(e, A, B);
      }
    }
#endif
  }
  SYNC();
  fs_finish_contacts(c);
  if (c.lane == 0 && scal[SC_NSLOT] > c.ly.ncon_max) scal[SC_NSLOT] = c.ly.ncon_max;
  SYNC();
  FS_CPROF(31);
}
