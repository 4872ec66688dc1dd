// wave_tree.h -- constraint pipeline of the wavefront-per-candidate kernels WITHOUT a stored constraint Jacobian
// (per-precision include, see rollout_wave.h). Covers models with ONE moving kinematic tree: elliptic, pyramidal or frictionless
// contacts against static geoms and between two bodies of the tree, limits of joints and of fixed tendons -- the Quadruped and the
// Humanoid of the BASELINE configs.
//
// Same primal problem and the same Newton iteration as wf_constraint_newton / oracle o_constraint_newton
// (1/2 |a - a_smooth|^2_M + s(J a - a_ref), exact line search, MuJoCo's termination tests), but every product with J is
// taken through the kinematic tree instead of through rows of a 64 x nv table:
//   * a contact on body b at point p has J = A S_b, with S_b = [cdof_k, k on the chain of b] (6 x chain) and A the
//     (condim x 6) map  row j < 3: [ (p - com) x f_j , f_j ],  row j >= 3: [ f_{j-3} , 0 ]  (f = contact frame);
//     J v = A (S_b v): one spatial vector per body, shared by all of its contacts; J' f = S_b' (A' f); a contact between two
//     moving bodies has J = A (S_b2 - S_b1) (the dofs common to both chains cancel exactly);
//   * the Newton Hessian  M + J' H J = sum_b S_b' (I_b + X_b) S_b  with X_b = sum_c A_c' Hc A_c (6 x 6): the composite
//     rigid body recursion with the contacts' curvature added to the bodies' inertias -- row i of H is
//     H_ij = M_ij + cdof_j . (sum_{c below i} X_c cdof_i), j on the chain of i;
//   * friction-loss and joint-limit rows have one non-zero entry: they live in the registers of the lane that owns
//     the dof / joint and touch H and the gradient on the diagonal only.
// Consequences: no efc_J (9.2 KB of LDS for the A1), no row assembly pass, no 64-row cap (limits and friction rows are
// not rows of a table any more; contacts are capped by the two lists below), and the Hessian costs O(nv * depth)
// instead of O(rows * depth^2).
namespace mjpcx { namespace WAVE_NS {

// Capacity of the two contact lists, per step. A rollout that overflows them is flagged (warning bit 32) and rolled out again by
// a second, small launch with the large lists (tree_kernel.h): the common case pays for 16 cones of LDS, no rollout fails for
// lack of space (MuJoCo grows its arena instead).
constexpr int kTreeMaxCone = 16, kTreeMaxSimple = 12;         // first pass (A1: every contact with the floor is a cone)
constexpr int kTreeMaxConeBig = 56, kTreeMaxSimpleBig = 32;   // second pass (one lane per contact: at most 64)

struct TreeData {
  // frictionless contacts: generator a = [off x n, n] of the single row, the body it acts on
  wreal* s_a;    // [kTreeMaxSimple][6]
  wreal* s_fd;   // [kTreeMaxSimple][2]: force of the row, D of the row if it is active (else 0); [0] holds dist until the rows are built
  int* s_body;   // [kTreeMaxSimple][2]: body of geom 1 (static: no dofs) and of geom 2
  int* s_geom;   // [kTreeMaxSimple][2]
  // elliptic cones
  wreal* c_geo;  // [cap_c][12]: off (3), frame (9)
  wreal* c_par;  // [cap_c][12]: mu, friction[5], D[6]
  wreal* c_jar;  // [cap_c][6]   ([0] holds dist until the rows are built)
  wreal* c_X;    // [cap_c][21]: A' Hc A, packed lower triangle, while the Hessian is assembled; otherwise [0..5] A' force =
                 //   [torque about the tree's com, force] and, until the Newton loop starts, [6..11] aref
  int* c_meta;   // [cap_c][5]: body of geom 1, body of geom 2, geom 1, geom 2, condim
  // cones beyond cap_c (a robot lying on the floor): records of kConeRec reals in GLOBAL memory, one slab per wavefront
  // (registered-model kernel only; nullptr: the list ends at cap_c). Slow, rare, and no rollout fails for lack of LDS.
  wreal* ovf;
  int cap_tot;   // cap_c, or cap_c + the slab's capacity
  wreal* Vb;     // [nbody][6]: S_b v for the vector at hand
  int* cnt;      // [0] simple contacts [1] cones
  int cap_s, cap_c;  // capacity of the two lists
};

__host__ __device__ inline size_t tree_lds_elems(int kTreeMaxSimple, int kTreeMaxCone) {  // (Vb lives elsewhere: wave_carve_tree)
  const size_t ints = (size_t)kTreeMaxSimple * 4 + (size_t)kTreeMaxCone * 5 + 2;
  return (size_t)kTreeMaxSimple * 8 + (size_t)kTreeMaxCone * (12 + 12 + 6 + 21) + (ints * sizeof(int) + sizeof(wreal) - 1) / sizeof(wreal) + 2;
}
constexpr int kConeRec = 56;                    // geo 12 | par 12 | jar 6 | X 21 | meta (5 ints) | pad
constexpr int kTreeMaxConeTotal = 64;           // one lane per cone
// view of one cone's storage (LDS arrays or a global record)
struct ConeRef { wreal *geo, *par, *jar, *X; int* meta; };
__device__ __forceinline__ TreeData tree_carve(wreal* p, int kTreeMaxSimple, int kTreeMaxCone) {
  TreeData t;
  t.cap_s = kTreeMaxSimple; t.cap_c = kTreeMaxCone; t.ovf = nullptr; t.cap_tot = kTreeMaxCone;
  auto take = [&](size_t n) { wreal* q = p; p += n; return q; };
  t.s_a = take(6 * kTreeMaxSimple); t.s_fd = take(2 * kTreeMaxSimple);
  t.c_geo = take(12 * kTreeMaxCone); t.c_par = take(12 * kTreeMaxCone); t.c_jar = take(6 * kTreeMaxCone);
  t.c_X = take(21 * kTreeMaxCone);
  t.Vb = nullptr;
  int* ip = reinterpret_cast<int*>(p);
  t.s_body = ip; t.s_geom = ip + 2 * kTreeMaxSimple; t.c_meta = ip + 4 * kTreeMaxSimple; t.cnt = t.c_meta + 5 * kTreeMaxCone;
  return t;
}
__device__ __forceinline__ ConeRef cone_lds(const TreeData& t, int i) { return ConeRef{t.c_geo + 12 * i, t.c_par + 12 * i, t.c_jar + 6 * i, t.c_X + 21 * i, t.c_meta + 5 * i}; }
__device__ __forceinline__ ConeRef cone_ovf(const TreeData& t, int i) {
  wreal* r = t.ovf + (size_t)kConeRec * (i - t.cap_c);
  return ConeRef{r, r + 12, r + 24, r + 30, reinterpret_cast<int*>(r + 51)};
}
// run a statement block on cone i through the view `c`: written once, compiled twice (LDS-typed / global-typed accesses)
#define WT_CONE(i, ...)                                                      \
  do {                                                                       \
    if ((i) < t.cap_c) { const ConeRef c = cone_lds(t, (i)); __VA_ARGS__ }   \
    else { const ConeRef c = cone_ovf(t, (i)); __VA_ARGS__ }                 \
  } while (0)

// rows that live in the registers of one lane
struct TreeRows {
  // friction loss, lane = dof
  bool f_on; wreal f_D, f_R, f_fl, f_aref, f_jar, f_force; int f_zone;
  // joint limits, lane = joint; side 0: lower bound (J = +1 on the dof), side 1: upper bound (J = -1)
  bool l_on[2]; wreal l_D[2], l_aref[2], l_jar[2], l_force[2]; int l_zone[2];
  int l_dof;         // dof of this lane's joint
  int jnt_of_dof;    // lane = dof: the (slide / hinge) joint that owns it, -1 for free / ball dofs
  // frictionless contact, lane = index in the list
  bool s_on; wreal s_D, s_aref, s_jar, s_force; int s_zone;
  unsigned s_mp, s_mm;  // dofs with J = +A cdof (chain of body 2 only) / J = -A cdof (chain of body 1 only); other lanes read them with v_readlane
  int s_pair, c_pair;   // both bodies move (body 1 may be an ANCESTOR of body 2: s_mm == 0, yet the dofs above body 1 cancel and must stay out of the Hessian rows)
  // limits of fixed tendons, lane = tendon; side 0: lower bound (J = +coef), side 1: upper bound (J = -coef); the Jacobian is the
  // tendon's wrap coefficients on the dofs of its joints (wt_tendon_coef)
  bool t_on[2]; wreal t_D[2], t_aref[2], t_jar[2], t_force[2]; int t_zone[2];
  // cone, lane = index in the list (its rows are in LDS). Pyramidal cones (m.cone != 1) keep the same record: par[6] is the D of
  // every edge, c_zone is (mask of active edges) << 2 (0 = none: 'top'), c_Dm is unused
  bool c_on; int c_dim, c_zone;
  wreal c_Dm;  // D_normal / (mu^2 (1 + mu^2)): the cone's middle-zone stiffness (one division per step, not per line-search trial)
  unsigned c_mp, c_mm;
};

// packed lower triangle helpers (i >= j at i (i + 1) / 2 + j)
__device__ __forceinline__ void sym6_rank1(wreal* X, wreal c, const wreal* a) {
#pragma unroll
  for (int i = 0; i < 6; i++) {
    const wreal ca = c * a[i];
#pragma unroll
    for (int j = 0; j <= i; j++) X[i * (i + 1) / 2 + j] += ca * a[j];
  }
}
__device__ __forceinline__ void sym3_rank1(wreal* X, wreal c, const wreal* a) {  // rotational block only
#pragma unroll
  for (int i = 0; i < 3; i++) {
    const wreal ca = c * a[i];
#pragma unroll
    for (int j = 0; j <= i; j++) X[i * (i + 1) / 2 + j] += ca * a[j];
  }
}
// y = X v, X packed symmetric 6 x 6 read from LDS
__device__ __forceinline__ void sym6_mulvec_acc(wreal* y, const wreal* X, const wreal* v) {
  wreal x[21];
#pragma unroll
  for (int q = 0; q < 21; q++) x[q] = X[q];
#pragma unroll
  for (int i = 0; i < 6; i++) {
    wreal s = 0;
#pragma unroll
    for (int j = 0; j < 6; j++) s += x[i >= j ? i * (i + 1) / 2 + j : j * (j + 1) / 2 + i] * v[j];
    y[i] += s;
  }
}

// contacts found by one pass of the narrow phase (<= K per lane, same geom pair per lane): split by class (condim of the
// pair, mj_contactParam) and appended to the two lists in lane-major, contact-minor order (= the oracle's detection order)
template <int K, class MODEL>
__device__ __forceinline__ void wt_emit(const MODEL& m, WaveData& d, TreeData& t, int lane, int cnt, const wreal* cd, const wreal (*cp)[3],
                                        const wreal (*cn)[3], int g1, int g2, int b1, int b2, const wreal* com) {
  int dim = 1;
  if (cnt > 0) {
    const int pr1 = m.geom_priority[g1], pr2 = m.geom_priority[g2];
    if (pr1 != pr2) dim = m.geom_condim[pr1 > pr2 ? g1 : g2];
    else dim = m.geom_condim[g1] > m.geom_condim[g2] ? m.geom_condim[g1] : m.geom_condim[g2];
  }
  const bool cone = dim > 1;
  int below_s = 0, total_s = 0, below_c = 0, total_c = 0;
  const unsigned long long lower = (1ull << lane) - 1ull;
#pragma unroll
  for (int k = 0; k < K; k++) {
    const unsigned long long bs = __ballot(cnt > k && !cone), bc = __ballot(cnt > k && cone);
    below_s += __popcll(bs & lower); total_s += __popcll(bs);
    below_c += __popcll(bc & lower); total_c += __popcll(bc);
  }
  if (total_s + total_c == 0) return;  // wave-uniform
  const int base_s = __builtin_amdgcn_readfirstlane(t.cnt[0]), base_c = __builtin_amdgcn_readfirstlane(t.cnt[1]);
  WSYNC();
#pragma unroll
  for (int k = 0; k < K; k++) {
    if (k < cnt) {
      wreal frame[9];
      for (int e = 0; e < 3; e++) frame[e] = cn[k][e];
      w_make_frame(frame);
      const wreal off[3] = {cp[k][0] - com[0], cp[k][1] - com[1], cp[k][2] - com[2]};
      if (!cone) {
        const int at = base_s + below_s + k;
        if (at < t.cap_s) {
          wreal rot[3];
          cr3(rot, off, frame);  // off x n
          for (int e = 0; e < 3; e++) { t.s_a[6 * at + e] = rot[e]; t.s_a[6 * at + 3 + e] = frame[e]; }
          t.s_fd[2 * at] = cd[k];
          t.s_body[2 * at] = b1; t.s_body[2 * at + 1] = b2; t.s_geom[2 * at] = g1; t.s_geom[2 * at + 1] = g2;
        }
      } else {
        const int at = base_c + below_c + k;
        if (at < t.cap_tot)
          WT_CONE(at,
            for (int e = 0; e < 3; e++) c.geo[e] = off[e];
            for (int e = 0; e < 9; e++) c.geo[3 + e] = frame[e];
            c.jar[0] = cd[k];
            c.meta[0] = b1; c.meta[1] = b2; c.meta[2] = g1; c.meta[3] = g2; c.meta[4] = dim;);
      }
    }
  }
  if (lane == 0) {
    const int ns = base_s + total_s, nc = base_c + total_c;
    if (ns > t.cap_s || nc > t.cap_tot) d.counters[2] |= 32;
    t.cnt[0] = ns > t.cap_s ? t.cap_s : ns;
    t.cnt[1] = nc > t.cap_tot ? t.cap_tot : nc;
  }
  WSYNC();
}

// ---- collision: wf_collision's narrow phase (o_collision), contacts split by class and compacted in detection order
template <class MODEL>
__device__ __forceinline__ void wt_collision(const MODEL& m, WaveData& d, TreeData& t, int lane) {
  if (lane == 0) { t.cnt[0] = 0; t.cnt[1] = 0; }
  WSYNC();
  if (m.disableflags & (MJPCX_DSBL_CONSTRAINT | MJPCX_DSBL_CONTACT)) return;
  const bool have = lane < m.ndynamic_geom;
  const int g2 = have ? m.dynamic_geom[lane] : 0;
  wreal p2[3] = {0, 0, 0}, R2[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
  if (have) wf_geom_pose(m, d, g2, p2, R2);
  const int t2 = have ? m.geom_type[g2] : -1;
  const int b2 = have ? m.geom_bodyid[g2] : 0;
  const wreal s2[3] = {have ? m.geom_size[3 * g2] : 0, have ? m.geom_size[3 * g2 + 1] : 0, have ? m.geom_size[3 * g2 + 2] : 0};
  wreal com[3] = {0, 0, 0};
  if (have) for (int k = 0; k < 3; k++) com[k] = d.subtree_com[3 * m.body_rootid[b2] + k];
  for (int si = 0; si < m.nstatic_geom; si++) {
    const int g1 = m.static_geom[si], t1 = m.geom_type[g1];
    if (t1 != MJPCX_GEOM_PLANE && t1 != MJPCX_GEOM_SPHERE && t1 != MJPCX_GEOM_BOX) continue;
    wreal p1[3], R1[9];
    wf_geom_pose(m, d, g1, p1, R1);  // wave-uniform
    wreal cd[4], cp[4][3], cn[3] = {0, 0, 1};
    int cnt = 0;
    auto push = [&](wreal dist, wreal px, wreal py, wreal pz) {
      switch (cnt) {
        case 0: cd[0] = dist; cp[0][0] = px; cp[0][1] = py; cp[0][2] = pz; break;
        case 1: cd[1] = dist; cp[1][0] = px; cp[1][1] = py; cp[1][2] = pz; break;
        case 2: cd[2] = dist; cp[2][0] = px; cp[2][1] = py; cp[2][2] = pz; break;
        default: cd[3] = dist; cp[3][0] = px; cp[3][1] = py; cp[3][2] = pz; break;
      }
      cnt++;
    };
    wreal margin = 0;
    const bool pair = have && ((m.geom_contype[g1] & m.geom_conaffinity[g2]) || (m.geom_contype[g2] & m.geom_conaffinity[g1]));
    if (pair) {
      margin = fmax(m.geom_margin[g1], m.geom_margin[g2]);
      if (t1 == MJPCX_GEOM_PLANE) {
        const wreal n[3] = {R1[2], R1[5], R1[8]};
        for (int k = 0; k < 3; k++) cn[k] = n[k];
        auto sphere_plane = [&](const wreal* c, wreal r) {
          const wreal dist = (c[0] - p1[0]) * n[0] + (c[1] - p1[1]) * n[1] + (c[2] - p1[2]) * n[2] - r;
          if (dist < margin) push(dist, c[0] - n[0] * (r + WL(0.5) * dist), c[1] - n[1] * (r + WL(0.5) * dist), c[2] - n[2] * (r + WL(0.5) * dist));
        };
        if (t2 == MJPCX_GEOM_SPHERE) {
          sphere_plane(p2, s2[0]);
        } else if (t2 == MJPCX_GEOM_CAPSULE) {
          for (int sgn = -1; sgn <= 1; sgn += 2) {
            wreal c[3];
            for (int k = 0; k < 3; k++) c[k] = p2[k] + sgn * s2[1] * R2[3 * k + 2];
            sphere_plane(c, s2[0]);
          }
        } else if (t2 == MJPCX_GEOM_BOX) {
          for (int i = 0; i < 8 && cnt < 4; i++) {
            const wreal loc[3] = {(i & 1 ? s2[0] : -s2[0]), (i & 2 ? s2[1] : -s2[1]), (i & 4 ? s2[2] : -s2[2])};
            wreal c[3];
            mv3(c, R2, loc);
            for (int k = 0; k < 3; k++) c[k] += p2[k];
            const wreal dist = (c[0] - p1[0]) * n[0] + (c[1] - p1[1]) * n[1] + (c[2] - p1[2]) * n[2];
            if (dist < margin) push(dist, c[0] - WL(0.5) * dist * n[0], c[1] - WL(0.5) * dist * n[1], c[2] - WL(0.5) * dist * n[2]);
          }
        } else if (t2 == MJPCX_GEOM_CYLINDER) {
          const wreal a[3] = {R2[2], R2[5], R2[8]};
          const wreal pa = n[0] * a[0] + n[1] * a[1] + n[2] * a[2];
          const wreal sgn = pa > 0 ? -WL(1.0) : WL(1.0);
          wreal v[3], vn = 0;
          for (int k = 0; k < 3; k++) { v[k] = -(n[k] - pa * a[k]); vn += v[k] * v[k]; }
          vn = sqrt(vn);
          if (vn < WL(1e-10)) { v[0] = R2[0]; v[1] = R2[3]; v[2] = R2[6]; vn = 1; }
          for (int k = 0; k < 3; k++) v[k] /= vn;
          wreal w[3];
          cr3(w, a, v);
          const wreal cs[3] = {WL(1.0), -WL(0.5), -WL(0.5)}, sn[3] = {WL(0.0), WL(0.8660254037844386), -WL(0.8660254037844386)};
          for (int i = 0; i < 4; i++) {
            const wreal side = i < 3 ? sgn : -sgn, cc = i < 3 ? cs[i] : WL(1.0), ss = i < 3 ? sn[i] : WL(0.0);
            wreal c[3];
            for (int k = 0; k < 3; k++) c[k] = p2[k] + side * s2[1] * a[k] + s2[0] * (cc * v[k] + ss * w[k]);
            const wreal dist = (c[0] - p1[0]) * n[0] + (c[1] - p1[1]) * n[1] + (c[2] - p1[2]) * n[2];
            if (dist < margin) push(dist, c[0] - WL(0.5) * dist * n[0], c[1] - WL(0.5) * dist * n[1], c[2] - WL(0.5) * dist * n[2]);
          }
        }
      } else if (t1 == MJPCX_GEOM_SPHERE && t2 == MJPCX_GEOM_SPHERE) {
        wreal n[3], len = 0;
        for (int k = 0; k < 3; k++) { n[k] = p2[k] - p1[k]; len += n[k] * n[k]; }
        len = sqrt(len);
        if (len < kMinVal) { n[0] = 1; n[1] = n[2] = 0; } else for (int k = 0; k < 3; k++) n[k] /= len;
        const wreal r1 = m.geom_size[3 * g1], dist = len - r1 - s2[0];
        if (dist < margin) {
          cd[0] = dist;
          for (int k = 0; k < 3; k++) { cp[0][k] = p1[k] + n[k] * (r1 + WL(0.5) * dist); cn[k] = n[k]; }
          cnt = 1;
        }
      } else if (t1 == MJPCX_GEOM_BOX && t2 == MJPCX_GEOM_SPHERE) {
        const wreal* s1 = m.geom_size + 3 * g1;
        wreal rel[3], loc[3], clamped[3];
        for (int k = 0; k < 3; k++) rel[k] = p2[k] - p1[k];
        for (int k = 0; k < 3; k++) loc[k] = R1[k] * rel[0] + R1[3 + k] * rel[1] + R1[6 + k] * rel[2];
        bool inside = true;
        for (int k = 0; k < 3; k++) {
          clamped[k] = loc[k] < -s1[k] ? -s1[k] : (loc[k] > s1[k] ? s1[k] : loc[k]);
          if (clamped[k] != loc[k]) inside = false;
        }
        wreal nl[3] = {0, 0, 0}, dist;
        if (!inside) {
          wreal len = 0;
          for (int k = 0; k < 3; k++) { nl[k] = loc[k] - clamped[k]; len += nl[k] * nl[k]; }
          len = sqrt(len);
          for (int k = 0; k < 3; k++) nl[k] /= len;
          dist = len - s2[0];
        } else {
          int best = 0; wreal bd = WL(1e300);
          for (int k = 0; k < 3; k++) { const wreal dd = s1[k] - fabs(loc[k]); if (dd < bd) { bd = dd; best = k; } }
          nl[best] = loc[best] >= 0 ? 1 : -1;
          clamped[best] = nl[best] * s1[best];
          dist = -bd - s2[0];
        }
        if (dist < margin) {
          wreal n[3], surf[3];
          mv3(n, R1, nl);
          mv3(surf, R1, clamped);
          cd[0] = dist;
          for (int k = 0; k < 3; k++) { cp[0][k] = p1[k] + surf[k] + WL(0.5) * dist * n[k]; cn[k] = n[k]; }
          cnt = 1;
        }
      }
    }
    wreal cnk[4][3];
    for (int k = 0; k < 4; k++) for (int e = 0; e < 3; e++) cnk[k][e] = cn[e];
    wt_emit<4>(m, d, t, lane, cnt, cd, cp, cnk, g1, g2, m.geom_bodyid[g1], b2, com);
  }
  // moving-geom pairs (sphere | capsule pairs, sphere | capsule against box | cylinder; oracle pair_collide): one lane per baked pair, up to two contacts each
  for (int p0 = 0; p0 < m.npair; p0 += 64) {
    const bool on = p0 + lane < m.npair;
    const int g1 = on ? m.pair_g1[p0 + lane] : 0, g2p = on ? m.pair_g2[p0 + lane] : 0;
    wreal cd[2] = {0, 0}, cp[2][3] = {{0, 0, 0}, {0, 0, 0}}, cn[2][3] = {{1, 0, 0}, {1, 0, 0}};
    int cnt = 0;
    bool near = false, solids_touch = false;
    wreal p1[3] = {0, 0, 0}, q2[3] = {0, 0, 0};
    const wreal margin = on ? fmax(m.geom_margin[g1], m.geom_margin[g2p]) : WL(0.0);
    const int pb1 = on ? m.geom_bodyid[g1] : 0, pb2 = on ? m.geom_bodyid[g2p] : 0;
    if (on) {
      wreal v1[3], v2[3];
      mv3(v1, d.xmat + 9 * pb1, m.geom_pos + 3 * g1);
      mv3(v2, d.xmat + 9 * pb2, m.geom_pos + 3 * g2p);
      wreal dd = 0;
      for (int k = 0; k < 3; k++) { p1[k] = d.xpos[3 * pb1 + k] + v1[k]; q2[k] = d.xpos[3 * pb2 + k] + v2[k]; dd += (p1[k] - q2[k]) * (p1[k] - q2[k]); }
      const wreal reach = wf_pair_bound(m, g1) + wf_pair_bound(m, g2p) + margin + WL(1e-6);
      near = dd <= reach * reach;
    }
    if (__ballot(near) == 0ull) continue;
    if (near) {
      wreal R1[9], R2p[9];
      wf_geom_pose(m, d, g1, p1, R1);
      wf_geom_pose(m, d, g2p, q2, R2p);
      const int t1 = m.geom_type[g1];
      const wreal r1 = m.geom_size[3 * g1], r2 = m.geom_size[3 * g2p];
      auto spheres = [&](const wreal* c1, const wreal* c2) {
        wreal n[3], len = 0;
        for (int k = 0; k < 3; k++) { n[k] = c2[k] - c1[k]; len += n[k] * n[k]; }
        len = sqrt(len);
        if (len < kMinVal) { n[0] = 1; n[1] = n[2] = 0; } else for (int k = 0; k < 3; k++) n[k] /= len;
        const wreal dist = len - r1 - r2;
        if (dist < margin) {
          if (cnt == 0) { cd[0] = dist; for (int k = 0; k < 3; k++) { cp[0][k] = c1[k] + n[k] * (r1 + WL(0.5) * dist); cn[0][k] = n[k]; } }
          else { cd[1] = dist; for (int k = 0; k < 3; k++) { cp[1][k] = c1[k] + n[k] * (r1 + WL(0.5) * dist); cn[1][k] = n[k]; } }
          cnt++;
        }
      };
      auto seg = [&](const wreal* p, const wreal* a, wreal h, const wreal* c) {
        const wreal x = (c[0] - p[0]) * a[0] + (c[1] - p[1]) * a[1] + (c[2] - p[2]) * a[2];
        return x < -h ? -h : (x > h ? h : x);
      };
      const int t2p = m.geom_type[g2p];
      if (t2p == MJPCX_GEOM_CYLINDER || t2p == MJPCX_GEOM_BOX) {  // (sphere | capsule, box | cylinder): solid_pairs.h; two solids are only watched
        cnt = wf_thin_vs_solid(m, g1, g2p, p1, R1, q2, R2p, margin, cd, cp[0], cn[0], t1 == MJPCX_GEOM_CYLINDER || t1 == MJPCX_GEOM_BOX);
        if (cnt < 0) { cnt = 0; solids_touch = true; }
      } else if (t1 == MJPCX_GEOM_SPHERE && t2p == MJPCX_GEOM_SPHERE) {
        spheres(p1, q2);
      } else if (t1 == MJPCX_GEOM_SPHERE) {
        const wreal a2[3] = {R2p[2], R2p[5], R2p[8]};
        const wreal x = seg(q2, a2, m.geom_size[3 * g2p + 1], p1);
        const wreal c2[3] = {q2[0] + x * a2[0], q2[1] + x * a2[1], q2[2] + x * a2[2]};
        spheres(p1, c2);
      } else {
        const wreal a1[3] = {R1[2], R1[5], R1[8]}, a2[3] = {R2p[2], R2p[5], R2p[8]};
        const wreal h1 = m.geom_size[3 * g1 + 1], h2 = m.geom_size[3 * g2p + 1];
        const wreal dif[3] = {p1[0] - q2[0], p1[1] - q2[1], p1[2] - q2[2]};
        const wreal mb = -(a1[0] * a2[0] + a1[1] * a2[1] + a1[2] * a2[2]);
        const wreal u = -(a1[0] * dif[0] + a1[1] * dif[1] + a1[2] * dif[2]);
        const wreal v = a2[0] * dif[0] + a2[1] * dif[1] + a2[2] * dif[2];
        const wreal det = WL(1.0) - mb * mb;
        wreal c1[3], c2[3];
        if (fabs(det) >= kMinVal) {
          wreal x1 = (u - mb * v) / det, x2 = (v - mb * u) / det;
          if (x1 > h1) { x1 = h1; x2 = v - mb * x1; } else if (x1 < -h1) { x1 = -h1; x2 = v - mb * x1; }
          if (x2 > h2) { x2 = h2; x1 = u - mb * x2; x1 = x1 > h1 ? h1 : (x1 < -h1 ? -h1 : x1); }
          else if (x2 < -h2) { x2 = -h2; x1 = u - mb * x2; x1 = x1 > h1 ? h1 : (x1 < -h1 ? -h1 : x1); }
          for (int k = 0; k < 3; k++) { c1[k] = p1[k] + x1 * a1[k]; c2[k] = q2[k] + x2 * a2[k]; }
          spheres(c1, c2);
        } else {
          for (int e = 0; e < 4 && cnt < 2; e++) {
            const wreal sgn = (e & 1) ? -WL(1.0) : WL(1.0);
            if (e < 2) {
              for (int k = 0; k < 3; k++) c1[k] = p1[k] + sgn * h1 * a1[k];
              const wreal x2 = seg(q2, a2, h2, c1);
              for (int k = 0; k < 3; k++) c2[k] = q2[k] + x2 * a2[k];
            } else {
              for (int k = 0; k < 3; k++) c2[k] = q2[k] + sgn * h2 * a2[k];
              const wreal x1 = seg(p1, a1, h1, c2);
              for (int k = 0; k < 3; k++) c1[k] = p1[k] + x1 * a1[k];
            }
            spheres(c1, c2);
          }
        }
      }
    }
    wreal pcom[3] = {0, 0, 0};
    if (on) for (int k = 0; k < 3; k++) pcom[k] = d.subtree_com[3 * m.body_rootid[pb2] + k];
    if (__ballot(solids_touch) != 0ull && lane == 0) d.counters[2] |= 128;  // (two solids within reach: no narrow phase -- the rollout fails, as the oracle's does)
    wt_emit<2>(m, d, t, lane, cnt, cd, cp, cn, g1, g2p, pb1, pb2, pcom);
  }
}

// ---- packed symmetric matrices (lower triangle, row-major: (i, j), i >= j, at i (i + 1) / 2 + j)
__device__ __forceinline__ int wt_tri(int i, int j) { return i >= j ? i * (i + 1) / 2 + j : j * (j + 1) / 2 + i; }
// (M v)[lane] for a packed symmetric M; lane < nv. (Entry b of v as a broadcast read of LDS: handing it round with v_readlane from the
// owning lane saves 2 nv LDS reads and is 1 % faster on the Humanoid, but the allocator then spills loop-carried solver state inside
// the Newton loop -- scratch traffic 12.6 -> 27 GB per launch, measured -- so the reads stay.)
template <int NMAX>
__device__ __forceinline__ wreal wt_sym_mulvec(const wreal* Mp, const wreal* v, int nv, int lane) {
  wreal s = 0;
  const int rowadr = lane * (lane + 1) / 2;
#pragma unroll
  for (int b = 0; b < NMAX; b++)
    if (b < nv) s += Mp[b <= lane ? rowadr + b : b * (b + 1) / 2 + lane] * v[b];
  return s;
}
// same with v = x - y
template <int NMAX>
__device__ __forceinline__ wreal wt_sym_mulvec_diff(const wreal* Mp, const wreal* x, const wreal* y, int nv, int lane) {
  wreal s = 0;
  const int rowadr = lane * (lane + 1) / 2;
#pragma unroll
  for (int b = 0; b < NMAX; b++)
    if (b < nv) s += Mp[b <= lane ? rowadr + b : b * (b + 1) / 2 + lane] * (x[b] - y[b]);
  return s;
}

// o_crb on the tree path: composite inertias by subtree masks, then M packed (entries off the kinematic chains are zero)
template <class MODEL>
__device__ __forceinline__ void wt_crb(const MODEL& m, WaveData& d, int lane) {
  const int nb = m.nbody, nv = m.nv;
  if (lane < nb && lane > 0) {
    const int i = lane;
    unsigned long long mask = m.body_subtree_mask[i];
    wreal s[10];
    for (int k = 0; k < 10; k++) s[k] = 0;
    while (mask) {
      const int j = __ffsll((long long)mask) - 1;
      mask &= mask - 1;
      for (int k = 0; k < 10; k++) s[k] += d.cinert[10 * j + k];
    }
    for (int k = 0; k < 10; k++) d.crb[10 * i + k] = s[k];
  }
  for (int e = lane; e < nv * (nv + 1) / 2; e += 64) d.M[e] = 0;
  WSYNC();
  if (lane < nv) {
    const int i = lane;
    wreal buf[6];
    w_mul_inert(buf, d.crb + 10 * m.dof_bodyid[i], d.cdof + 6 * i);
    d.M[wt_tri(i, i)] = m.dof_armature[i] + w_dot6(d.cdof + 6 * i, buf);
    for (int j = m.dof_parentid[i]; j >= 0; j = m.dof_parentid[j]) d.M[wt_tri(i, j)] = w_dot6(d.cdof + 6 * j, buf);
  }
  WSYNC();
}

// Cholesky of a packed SPD matrix, lane i owns row i in registers (values of other rows through v_readlane): dst := L with
// src = L L' (dst may be src), dinv[j] = 1 / L[j][j]. LDS-typed pointers: ds_read / ds_write, never FLAT.
#ifdef MJPCX_TREE_INLINE_CHOL
#define WT_CHOL_LINKAGE __forceinline__
#else
#define WT_CHOL_LINKAGE __noinline__
#endif
template <int NMAX>
__device__ WT_CHOL_LINKAGE bool wt_chol(const wlds_f64* src, wlds_f64* dst, wlds_f64* dinv, int n_, int lane) {
  const int n = __builtin_amdgcn_readfirstlane(n_);
  const int rowadr = lane * (lane + 1) / 2;
  wreal row[NMAX];
#pragma unroll
  for (int k = 0; k < NMAX; k++) row[k] = (lane < n && k <= lane) ? src[rowadr + k] : WL(0.0);
  bool ok = true;
#pragma unroll
  for (int j = 0; j < NMAX; j++) {
    if (j < n && ok) {
      const wreal djj = wbcast(row[j], j);
      if (!(djj > kMinVal)) {
        ok = false;
      } else {
        const wreal inv = rsqrt(djj);
        const wreal lij = lane == j ? djj * inv : row[j] * inv;
        row[j] = lij;
        if (lane == j) dinv[j] = inv;
#pragma unroll
        for (int k = j + 1; k < NMAX; k++) row[k] -= lij * wbcast(lij, k);
      }
    }
  }
#pragma unroll
  for (int k = 0; k < NMAX; k++) if (lane < n && k <= lane) dst[rowadr + k] = row[k];
  WSYNC();
  return ok;
}
// x := (L L')^-1 x, L packed, x in LDS
template <int NMAX>
__device__ WT_CHOL_LINKAGE void wt_chol_solve(wlds_f64* x, const wlds_f64* L, const wlds_f64* dinv, int n_, int lane) {
  const int n = __builtin_amdgcn_readfirstlane(n_);
  const int rowadr = lane * (lane + 1) / 2;
  wreal row[NMAX], col[NMAX];
#pragma unroll
  for (int k = 0; k < NMAX; k++) {
    row[k] = (lane < n && k < lane) ? L[rowadr + k] : WL(0.0);
    col[k] = (lane < n && k > lane && k < n) ? L[k * (k + 1) / 2 + lane] : WL(0.0);
  }
  wreal b = lane < n ? x[lane] : WL(0.0);
  const wreal mydinv = lane < n ? dinv[lane] : WL(0.0);
#pragma unroll
  for (int j = 0; j < NMAX; j++) {
    const wreal yj = wbcast(b, j) * wbcast(mydinv, j);
    b = lane == j ? yj : (lane > j ? b - row[j] * yj : b);
  }
#pragma unroll
  for (int j = NMAX - 1; j >= 0; j--) {
    const wreal xj = wbcast(b, j) * wbcast(mydinv, j);
    b = lane == j ? xj : (lane < j ? b - col[j] * xj : b);
  }
  if (lane < n) x[lane] = b;
  WSYNC();
}

// o_euler on the tree path (packed M): implicit joint damping, then integrate positions
template <int NMAX, class MODEL>
__device__ __forceinline__ void wt_euler(const MODEL& m, WaveData& d, int lane, wreal& time) {
  const int nv = m.nv;
  const wreal h = m.timestep;
  if (m.any_damping && !(m.disableflags & MJPCX_DSBL_EULERDAMP)) {
    for (int e = lane; e < nv * (nv + 1) / 2; e += 64) d.H[e] = d.M[e];
    WSYNC();
    if (lane < nv) { d.H[wt_tri(lane, lane)] += h * m.dof_damping[lane]; d.tmpv[lane] = d.qfrc_smooth[lane] + d.qfrc_constraint[lane]; }
    WSYNC();
    if (wt_chol<NMAX>((const wlds_f64*)d.H, (wlds_f64*)d.H, (wlds_f64*)d.dinv, nv, lane)) wt_chol_solve<NMAX>((wlds_f64*)d.tmpv, (const wlds_f64*)d.H, (const wlds_f64*)d.dinv, nv, lane);
    else { if (lane < nv) d.tmpv[lane] = d.qacc[lane]; WSYNC(); }
  } else {
    if (lane < nv) d.tmpv[lane] = d.qacc[lane];
    WSYNC();
  }
  if (lane < nv) d.qvel[lane] += h * d.tmpv[lane];
  WSYNC();
  w_integrate_pos(m, d, d.qvel, h, lane);
  time += h;
  WSYNC();
}

// rows of a cone from a spatial vector V = [rot, lin] of its body (about the tree's com): x_j = a_j . V
__device__ __forceinline__ void wt_cone_project(wreal* x, const wreal* geo, const wreal* V, int dim) {
  const wreal off[3] = {geo[0], geo[1], geo[2]};
  wreal lin[3];
  cr3(lin, V, off);  // rot x off
  for (int k = 0; k < 3; k++) lin[k] += V[3 + k];
#pragma unroll
  for (int j = 0; j < 6; j++) {
    const wreal* f = geo + 3 + 3 * (j < 3 ? j : j - 3);
    const wreal* w = j < 3 ? lin : V;
    x[j] = j < dim ? f[0] * w[0] + f[1] * w[1] + f[2] * w[2] : WL(0.0);
  }
}
// generators a_j of a cone's rows (6 values each)
__device__ __forceinline__ void wt_cone_gen(wreal* a, const wreal* geo, int j) {
  const wreal* f = geo + 3 + 3 * (j < 3 ? j : j - 3);
  if (j < 3) { cr3(a, geo, f); a[3] = f[0]; a[4] = f[1]; a[5] = f[2]; }
  else { a[0] = f[0]; a[1] = f[1]; a[2] = f[2]; a[3] = a[4] = a[5] = 0; }
}

// penalty of an elliptic cone at x (oracle constraint_cost, EFC_ELLIPTIC): cost, forces, zone
__device__ __forceinline__ wreal wt_cone_cost(const wreal* x, const wreal* par, wreal Dm, int dim, wreal* force, int& zone) {
  const wreal mu = par[0];
  wreal U[6], T = 0, cost = 0;
  U[0] = x[0] * mu;
#pragma unroll
  for (int j = 1; j < 6; j++) { U[j] = j < dim ? x[j] * par[j] : WL(0.0); T += U[j] * U[j]; }
  T = sqrt(T);
  const wreal N = U[0];
  if (N >= mu * T || (T <= 0 && N >= 0)) {
#pragma unroll
    for (int j = 0; j < 6; j++) force[j] = 0;
    zone = kZoneTop;
  } else if (mu * N + T <= 0 || (T <= 0 && N < 0)) {
#pragma unroll
    for (int j = 0; j < 6; j++) {
      const wreal Dj = j < dim ? par[6 + j] : WL(0.0);
      cost += WL(0.5) * Dj * x[j] * x[j];
      force[j] = -Dj * x[j];
    }
    zone = kZoneBottom;
  } else {
    const wreal NT = N - mu * T;
    cost = WL(0.5) * Dm * NT * NT;
    force[0] = -Dm * NT * mu;
    const wreal kT = Dm * NT * mu / T;  // (one division; the oracle divides per row: a few ulp)
#pragma unroll
    for (int j = 1; j < 6; j++) force[j] = j < dim ? kT * U[j] * par[j] : WL(0.0);
    zone = kZoneMiddle;
  }
  return cost;
}
// first / second derivative along v at x (oracle constraint_line, EFC_ELLIPTIC)
__device__ __forceinline__ void wt_cone_line(const wreal* x0, const wreal* v, wreal alpha, const wreal* par, wreal Dm, int dim, wreal& g1, wreal& h2) {
  const wreal mu = par[0];
  wreal U[6], V[6], X[6], T = 0;
  X[0] = x0[0] + alpha * v[0]; U[0] = X[0] * mu; V[0] = v[0] * mu;
#pragma unroll
  for (int j = 1; j < 6; j++) {
    if (j < dim) { X[j] = x0[j] + alpha * v[j]; U[j] = X[j] * par[j]; V[j] = v[j] * par[j]; T += U[j] * U[j]; }
    else { X[j] = U[j] = V[j] = 0; }
  }
  T = sqrt(T);
  const wreal N = U[0];
  g1 = 0; h2 = 0;
  if (N >= mu * T || (T <= 0 && N >= 0)) {
  } else if (mu * N + T <= 0 || (T <= 0 && N < 0)) {
#pragma unroll
    for (int j = 0; j < 6; j++)
      if (j < dim) { g1 += par[6 + j] * X[j] * v[j]; h2 += par[6 + j] * v[j] * v[j]; }
  } else {
    const wreal NT = N - mu * T;
    wreal UV = 0, VV = 0;
#pragma unroll
    for (int j = 1; j < 6; j++) if (j < dim) { UV += U[j] * V[j]; VV += V[j] * V[j]; }
    const wreal iT = WL(1.0) / T;
    const wreal dNT = V[0] - mu * UV * iT;
    const wreal d2NT = -mu * (VV * iT - UV * UV * iT * iT * iT);
    g1 = Dm * NT * dNT;
    h2 = Dm * (dNT * dNT + NT * d2NT);
  }
}

// Pyramidal cone (oracle EFC_PYRAMID rows): 2 (dim - 1) edges  r = x_0 +- f_j x_j >= 0, each with the penalty 1/2 D min(0, r)^2.
// Same order as the oracle's rows: (j = 1, +), (j = 1, -), (j = 2, +), ...; bit 2 (j - 1) + (0: +, 1: -) of the mask = edge active.
__device__ __forceinline__ wreal wt_pyr_cost(const wreal* x, const wreal* par, int dim, wreal* force, int& zone) {
  const wreal D = par[6];
  wreal cost = 0;
  int mask = 0;
#pragma unroll
  for (int j = 0; j < 6; j++) force[j] = 0;
#pragma unroll
  for (int j = 1; j < 6; j++)
    if (j < dim) {
      const wreal f = par[j], rp = x[0] + f * x[j], rm = x[0] - f * x[j];
      if (rp < 0) { cost += WL(0.5) * D * rp * rp; force[0] -= D * rp; force[j] -= D * f * rp; mask |= 1 << (2 * (j - 1)); }
      if (rm < 0) { cost += WL(0.5) * D * rm * rm; force[0] -= D * rm; force[j] += D * f * rm; mask |= 2 << (2 * (j - 1)); }
    }
  zone = mask << 2;
  return cost;
}
__device__ __forceinline__ void wt_pyr_line(const wreal* x0, const wreal* v, wreal alpha, const wreal* par, int dim, wreal& g1, wreal& h2) {
  const wreal D = par[6], xn = x0[0] + alpha * v[0];
  g1 = 0; h2 = 0;
#pragma unroll
  for (int j = 1; j < 6; j++)
    if (j < dim) {
      const wreal f = par[j], xj = x0[j] + alpha * v[j];
      const wreal rp = xn + f * xj, rm = xn - f * xj, vp = v[0] + f * v[j], vm = v[0] - f * v[j];
      if (rp < 0) { g1 += D * rp * vp; h2 += D * vp * vp; }
      if (rm < 0) { g1 += D * rm * vm; h2 += D * vm * vm; }
    }
}
// coefficient of fixed tendon i on dof k (0 if its Jacobian does not touch the dof): the wrap list is two or three entries long
template <class MODEL>
__device__ __forceinline__ wreal wt_tendon_coef(const MODEL& m, int i, int k) {
  wreal c = 0;
  for (int w = m.tendon_adr[i]; w < m.tendon_adr[i] + m.tendon_num[i]; w++)
    if (m.jnt_dofadr[m.wrap_objid[w]] == k) c += m.wrap_prm[w];
  return c;
}
// J v of this lane's tendon (lane < ntendon) for a vector v in LDS
template <class MODEL>
__device__ __forceinline__ wreal wt_tendon_jv(const MODEL& m, int lane, const wreal* v) {
  wreal s = 0;
  for (int w = m.tendon_adr[lane]; w < m.tendon_adr[lane] + m.tendon_num[lane]; w++) s += m.wrap_prm[w] * v[m.jnt_dofadr[m.wrap_objid[w]]];
  return s;
}

// ---- rows: friction loss (lane = dof), joint limits (lane = joint), contacts (lane = list index); impedance, reference
// acceleration and regulariser per row as o_make_constraint_full. J qvel of a contact row is the body's cvel projected.
template <class MODEL>
__device__ __forceinline__ void wt_make_constraint(const MODEL& m, WaveData& d, TreeData& t, TreeRows& q, int lane) {
  const int nv = m.nv;
  q.f_on = false; q.l_on[0] = q.l_on[1] = false; q.s_on = false; q.c_on = false;
  q.f_D = q.f_R = q.f_fl = q.f_aref = q.f_jar = q.f_force = 0; q.f_zone = kZoneTop;
  q.s_D = q.s_aref = q.s_jar = q.s_force = 0; q.s_zone = kZoneTop; q.c_dim = 0; q.c_zone = kZoneTop; q.s_mp = q.s_mm = q.c_mp = q.c_mm = 0; q.c_Dm = 0;
  q.s_pair = q.c_pair = 0;
  for (int s = 0; s < 2; s++) { q.l_D[s] = q.l_aref[s] = q.l_jar[s] = q.l_force[s] = 0; q.l_zone[s] = kZoneTop; }
  for (int s = 0; s < 2; s++) { q.t_on[s] = false; q.t_D[s] = q.t_aref[s] = q.t_jar[s] = q.t_force[s] = 0; q.t_zone[s] = kZoneTop; }
  q.l_dof = 0; q.jnt_of_dof = -1;
  if (lane < nv) {
    const int j = m.dof_jntid[lane], jt = m.jnt_type[j];
    q.jnt_of_dof = (jt == kJntSlide || jt == kJntHinge) ? j : -1;
  }
  if (m.disableflags & MJPCX_DSBL_CONSTRAINT) { if (lane == 0) { t.cnt[0] = 0; t.cnt[1] = 0; } WSYNC(); return; }
  // friction loss
  if (!(m.disableflags & MJPCX_DSBL_FRICTIONLOSS) && lane < nv && m.dof_frictionloss[lane] > 0) {
    wreal kk, bb;
    w_solref_kb(m, m.dof_solref + 2 * lane, m.dof_solimp + 5 * lane, kk, bb);
    const wreal imp = w_impedance(m.dof_solimp + 5 * lane, WL(0.0));
    wreal R = (1 - imp) / imp * m.dof_invweight0[lane];
    if (R < kMinVal) R = kMinVal;
    q.f_on = true; q.f_R = R; q.f_D = WL(1.0) / R; q.f_fl = m.dof_frictionloss[lane];
    q.f_aref = -bb * d.qvel[lane] - kk * imp * WL(0.0);
  }
  // joint limits
  if (!(m.disableflags & MJPCX_DSBL_LIMIT) && lane < m.njnt && m.jnt_limited[lane] &&
      (m.jnt_type[lane] == kJntSlide || m.jnt_type[lane] == kJntHinge)) {
    const wreal value = d.qpos[m.jnt_qposadr[lane]], margin = m.jnt_margin[lane];
    const int dof = m.jnt_dofadr[lane];
    q.l_dof = dof;
    wreal kk, bb;
    w_solref_kb(m, m.jnt_solref + 2 * lane, m.jnt_solimp + 5 * lane, kk, bb);
#pragma unroll
    for (int s = 0; s < 2; s++) {
      const wreal dist = s == 0 ? -(m.jnt_range[2 * lane] - value) : m.jnt_range[2 * lane + 1] - value;
      if (dist < margin) {
        const wreal pos = dist - margin;
        const wreal imp = w_impedance(m.jnt_solimp + 5 * lane, pos);
        wreal R = (1 - imp) / imp * m.dof_invweight0[dof];
        if (R < kMinVal) R = kMinVal;
        const wreal vel = s == 0 ? d.qvel[dof] : -d.qvel[dof];
        q.l_on[s] = true; q.l_D[s] = WL(1.0) / R; q.l_aref[s] = -bb * vel - kk * imp * pos;
      }
    }
  }
  // limits of fixed tendons (o_make_constraint_full, EFC_TENDON): length = sum coef q_joint
  if (!(m.disableflags & MJPCX_DSBL_LIMIT) && lane < m.ntendon && m.tendon_limited[lane]) {
    wreal value = 0;
    for (int w = m.tendon_adr[lane]; w < m.tendon_adr[lane] + m.tendon_num[lane]; w++) value += m.wrap_prm[w] * d.qpos[m.jnt_qposadr[m.wrap_objid[w]]];
    const wreal margin = m.tendon_margin[lane], jv = wt_tendon_jv(m, lane, d.qvel);
    wreal kk, bb;
    w_solref_kb(m, m.tendon_solref_lim + 2 * lane, m.tendon_solimp_lim + 5 * lane, kk, bb);
#pragma unroll
    for (int s = 0; s < 2; s++) {
      const wreal dist = s == 0 ? -(m.tendon_range[2 * lane] - value) : m.tendon_range[2 * lane + 1] - value;
      if (dist < margin) {
        const wreal pos = dist - margin;
        const wreal imp = w_impedance(m.tendon_solimp_lim + 5 * lane, pos);
        wreal R = (1 - imp) / imp * m.tendon_invweight0[lane];
        if (R < kMinVal) R = kMinVal;
        const wreal vel = s == 0 ? jv : -jv;
        q.t_on[s] = true; q.t_D[s] = WL(1.0) / R; q.t_aref[s] = -bb * vel - kk * imp * pos;
      }
    }
  }
  const int ns = __builtin_amdgcn_readfirstlane(t.cnt[0]), nc = __builtin_amdgcn_readfirstlane(t.cnt[1]);
  // frictionless contacts
  if (lane < ns) {
    const int g1 = t.s_geom[2 * lane], g2 = t.s_geom[2 * lane + 1], b1 = t.s_body[2 * lane], b2 = t.s_body[2 * lane + 1];
    WaveContact c;
    wf_contact_param(m, g1, g2, c);
    const wreal dist = t.s_fd[2 * lane];
    wreal kk, bb;
    w_solref_kb(m, c.solref, c.solimp, kk, bb);
    const wreal pos = dist - c.includemargin;
    const wreal imp = w_impedance(c.solimp, pos);
    const wreal diag = m.body_invweight0[2 * b1] + m.body_invweight0[2 * b2];
    wreal R = (1 - imp) / imp * diag;
    if (R < kMinVal) R = kMinVal;
    wreal vel = 0;
    for (int e = 0; e < 6; e++) vel += t.s_a[6 * lane + e] * (d.cvel[6 * b2 + e] - d.cvel[6 * b1 + e]);
    q.s_on = true; q.s_D = WL(1.0) / R; q.s_aref = -bb * vel - kk * imp * pos;
    { const unsigned m1 = m.body_dofmask[b1], m2 = m.body_dofmask[b2]; q.s_mp = m2 & ~m1; q.s_mm = m1 & ~m2; q.s_pair = (m1 != 0 && m2 != 0) ? 1 : 0; }
  }
  // elliptic cones
  if (lane < nc)
    WT_CONE(lane,
      const int b1 = c.meta[0]; const int b2 = c.meta[1]; const int g1 = c.meta[2]; const int g2 = c.meta[3]; const int dim = c.meta[4];
      WaveContact cp;
      wf_contact_param(m, g1, g2, cp);
      const wreal dist = c.jar[0];
      wreal kk; wreal bb;
      w_solref_kb(m, cp.solref, cp.solimp, kk, bb);
      const wreal pos = dist - cp.includemargin;
      const wreal imp = w_impedance(cp.solimp, pos);
      const wreal diag = m.body_invweight0[2 * b1] + m.body_invweight0[2 * b2];
      wreal R0 = (1 - imp) / imp * diag;
      if (R0 < kMinVal) R0 = kMinVal;
      const bool pyr = m.cone != 1;
      const wreal mu = pyr ? cp.friction[0] : cp.friction[0] / sqrt(m.impratio > kMinVal ? (wreal)m.impratio : WL(1.0));
      wreal Dpy = 0;
      if (pyr) {  // every edge gets Rpy = 2 mu^2 R of the first edge, whose mj_diagApprox is tran + friction_0^2 tran (mj_makeImpedance)
        const wreal f0 = cp.friction[0];
        wreal R1 = (1 - imp) / imp * (diag + f0 * f0 * diag);
        if (R1 < kMinVal) R1 = kMinVal;
        wreal Rpy = 2 * mu * mu * R1;
        if (Rpy < kMinVal) Rpy = kMinVal;
        Dpy = WL(1.0) / Rpy;
      }
      wreal geo[12]; wreal vel[6]; wreal cv[6];
      for (int e = 0; e < 12; e++) geo[e] = c.geo[e];
      for (int e = 0; e < 6; e++) cv[e] = d.cvel[6 * b2 + e] - d.cvel[6 * b1 + e];
      wt_cone_project(vel, geo, cv, dim);
      wreal* par = c.par;
      wreal* aref = c.X + 6;
      par[0] = mu;
      _Pragma("unroll")
      for (int j = 0; j < 6; j++) {
        if (j >= 1) par[j] = cp.friction[j - 1];
        wreal Rj = R0;
        if (j >= 1) { const wreal f = cp.friction[j - 1]; Rj = R0 * (mu * mu) / (f * f); }
        par[6 + j] = j < dim ? (pyr ? Dpy : WL(1.0) / Rj) : WL(0.0);
        aref[j] = j == 0 ? -bb * vel[0] - kk * imp * pos : -bb * vel[j];
      }
      q.c_on = true; q.c_dim = dim; q.c_Dm = pyr ? WL(0.0) : par[6] / (mu * mu * (1 + mu * mu));
      { const unsigned m1 = m.body_dofmask[b1]; const unsigned m2 = m.body_dofmask[b2]; q.c_mp = m2 & ~m1; q.c_mm = m1 & ~m2; q.c_pair = (m1 != 0 && m2 != 0) ? 1 : 0; });
  WSYNC();
}

// Vb[b] = sum over the dofs k on the chain of body b of cdof_k v[k]  (one lane per body)
template <class MODEL>
__device__ __forceinline__ void wt_body_vectors(const MODEL& m, const WaveData& d, TreeData& t, const wreal* v, int lane) {
  if (lane < m.nbody) {
    unsigned mask = m.body_dofmask[lane];
    wreal V[6] = {0, 0, 0, 0, 0, 0};
    while (mask) {
      const int k = __ffs((int)mask) - 1;
      mask &= mask - 1;
      const wreal vk = v[k];
      for (int e = 0; e < 6; e++) V[e] += d.cdof[6 * k + e] * vk;
    }
    for (int e = 0; e < 6; e++) t.Vb[6 * lane + e] = V[e];
  }
  WSYNC();
}

// value of this lane's limit rows pulled to the lane of their dof (lane = joint -> lane = dof); 0 for dofs without a joint row
__device__ __forceinline__ wreal wt_pull_from_joint(wreal v_on_joint_lane, int jnt_of_dof) {
  const wreal got = __shfl(v_on_joint_lane, jnt_of_dof < 0 ? 0 : jnt_of_dof, 64);
  return jnt_of_dof < 0 ? WL(0.0) : got;
}

// cost of all rows at the current jar (registers / c_jar); writes forces and zones (registers, s_fd, c_F). Wave-uniform sum.
__device__ __forceinline__ wreal wt_cost(WaveData& d, TreeData& t, TreeRows& q, int lane, bool pyr) {
  wreal cost = 0;
  if (q.f_on) {
    const wreal x = q.f_jar, f = q.f_fl, R = q.f_R, D = q.f_D;
    if (x <= -R * f) { cost += -WL(0.5) * R * f * f - f * x; q.f_force = f; q.f_zone = kZoneTop; }
    else if (x >= R * f) { cost += -WL(0.5) * R * f * f + f * x; q.f_force = -f; q.f_zone = kZoneTop; }
    else { cost += WL(0.5) * D * x * x; q.f_force = -D * x; q.f_zone = kZoneBottom; }
  }
#pragma unroll
  for (int s = 0; s < 2; s++)
    if (q.l_on[s]) {
      const wreal x = q.l_jar[s], D = q.l_D[s];
      if (x < 0) { cost += WL(0.5) * D * x * x; q.l_force[s] = -D * x; q.l_zone[s] = kZoneBottom; }
      else { q.l_force[s] = 0; q.l_zone[s] = kZoneTop; }
    }
#pragma unroll
  for (int s = 0; s < 2; s++)
    if (q.t_on[s]) {
      const wreal x = q.t_jar[s], D = q.t_D[s];
      if (x < 0) { cost += WL(0.5) * D * x * x; q.t_force[s] = -D * x; q.t_zone[s] = kZoneBottom; }
      else { q.t_force[s] = 0; q.t_zone[s] = kZoneTop; }
    }
  if (q.s_on) {
    const wreal x = q.s_jar, D = q.s_D;
    if (x < 0) { cost += WL(0.5) * D * x * x; q.s_force = -D * x; q.s_zone = kZoneBottom; }
    else { q.s_force = 0; q.s_zone = kZoneTop; }
    t.s_fd[2 * lane] = q.s_force;
    t.s_fd[2 * lane + 1] = q.s_zone == kZoneBottom ? D : WL(0.0);
  }
  if (q.c_on)
    WT_CONE(lane,
      wreal x[6]; wreal par[12]; wreal force[6]; wreal geo[12];
      for (int e = 0; e < 6; e++) x[e] = c.jar[e];
      for (int e = 0; e < 12; e++) par[e] = c.par[e];
      for (int e = 0; e < 12; e++) geo[e] = c.geo[e];
      cost += pyr ? wt_pyr_cost(x, par, q.c_dim, force, q.c_zone) : wt_cone_cost(x, par, q.c_Dm, q.c_dim, force, q.c_zone);
      // A' force = [off x F + sum_{j>=3} f_{j-3} force_j, F], F = sum_{j<3} f_j force_j
      wreal F[3] = {0, 0, 0}; wreal tq[3] = {0, 0, 0};
      _Pragma("unroll")
      for (int j = 0; j < 3; j++)
        for (int e = 0; e < 3; e++) { F[e] += geo[3 + 3 * j + e] * force[j]; tq[e] += geo[3 + 3 * j + e] * force[3 + j]; }
      wreal cx[3];
      cr3(cx, geo, F);
      for (int e = 0; e < 3; e++) { c.X[e] = cx[e] + tq[e]; c.X[3 + e] = F[e]; });
  cost = wave_sum(cost);
  WSYNC();
  return cost;
}

// sign of dof k in J = A (S_b2 - S_b1): +1 on the chain of body 2 only, -1 on the chain of body 1 only, 0 elsewhere
template <class MODEL>
__device__ __forceinline__ int wt_sign(const MODEL& m, int b1, int b2, int k) {
  return (int)((m.body_dofmask[b2] >> k) & 1u) - (int)((m.body_dofmask[b1] >> k) & 1u);
}

// (J' force)[dof = lane] from the forces written by wt_cost
template <class MODEL>
__device__ __forceinline__ wreal wt_jt_force(const MODEL& m, const WaveData& d, const TreeData& t, const TreeRows& q, int ns, int nc, int lane) {
  wreal s = q.f_on ? q.f_force : WL(0.0);
  const wreal lim = (q.l_on[0] ? q.l_force[0] : WL(0.0)) - (q.l_on[1] ? q.l_force[1] : WL(0.0));
  s += wt_pull_from_joint(lim, q.jnt_of_dof);
  for (int i = 0; i < m.ntendon; i++) {  // (wave-uniform trip count; the force of tendon i comes from its lane's registers)
    const wreal fi = (wreal)__shfl((q.t_on[0] ? q.t_force[0] : WL(0.0)) - (q.t_on[1] ? q.t_force[1] : WL(0.0)), i, 64);
    if (fi != 0 && lane < m.nv && ((m.tendon_dofmask[i] >> lane) & 1u)) s += wt_tendon_coef(m, i, lane) * fi;
  }
  if (lane < m.nv) {
    wreal cd[6];
    for (int e = 0; e < 6; e++) cd[e] = d.cdof[6 * lane + e];
    // contact i's dof masks come from its lane's registers (v_readlane, i is wave-uniform): no dependent LDS round trips
    for (int i = 0; i < ns; i++) {
      const unsigned mp = (unsigned)__builtin_amdgcn_readlane((int)q.s_mp, i), mm = (unsigned)__builtin_amdgcn_readlane((int)q.s_mm, i);
      const int sg = (int)((mp >> lane) & 1u) - (int)((mm >> lane) & 1u);
      if (sg != 0) {
        wreal ja = 0;
        for (int e = 0; e < 6; e++) ja += t.s_a[6 * i + e] * cd[e];
        s += (sg > 0 ? ja : -ja) * t.s_fd[2 * i];
      }
    }
    for (int i = 0; i < nc; i++) {
      const unsigned mp = (unsigned)__builtin_amdgcn_readlane((int)q.c_mp, i), mm = (unsigned)__builtin_amdgcn_readlane((int)q.c_mm, i);
      const int sg = (int)((mp >> lane) & 1u) - (int)((mm >> lane) & 1u);
      if (sg != 0)
        WT_CONE(i,
          wreal ja = 0;
          for (int e = 0; e < 6; e++) ja += c.X[e] * cd[e];
          s += sg > 0 ? ja : -ja;);
    }
  }
  return s;
}

// rows of a contact between two moving bodies: H[i][j] += s_i s_j cdof_j . zc over the dofs j <= i the contact touches (zc = X cdof_i)
__device__ __forceinline__ void wt_pair_rows(WaveData& d, int lane, const wreal* zc, unsigned mp, unsigned mm, int sg, unsigned lowmask) {
  unsigned both = (mp | mm) & lowmask;
  while (both) {
    const int j = __ffs((int)both) - 1;
    both &= both - 1;
    wreal s = 0;
    for (int e = 0; e < 6; e++) s += d.cdof[6 * j + e] * zc[e];
    const int sj = ((int)((mp >> j) & 1u) - (int)((mm >> j) & 1u)) * sg;
    d.H[wt_tri(lane, j)] += sj > 0 ? s : -s;
  }
}

// ---- the Newton solver (o_constraint_newton) on the tree
template <int NMAX, class MODEL>
__device__ __forceinline__ void wt_constraint_newton(const MODEL& m, WaveData& d, TreeData& t, TreeRows& q, int lane, long long* stamp = nullptr,
                                                     bool have_warm = false) {
  const int nv = m.nv;
  const int ns = __builtin_amdgcn_readfirstlane(t.cnt[0]), nc = __builtin_amdgcn_readfirstlane(t.cnt[1]);
  if (lane < nv) { d.qfrc_constraint[lane] = 0; d.qacc[lane] = d.qacc_smooth[lane]; }
  WSYNC();
  const bool pyr = m.cone != 1;
  const bool any_row = __any(q.f_on || q.l_on[0] || q.l_on[1] || q.t_on[0] || q.t_on[1] || q.s_on || q.c_on);
  if (!any_row) return;
  // jar = J qacc - aref at a vector v (LDS, nv entries): registers for the lane-owned rows, c_jar for the cones
  auto set_jar = [&](const wreal* v) {
    wt_body_vectors(m, d, t, v, lane);
    if (q.f_on) q.f_jar = v[lane] - q.f_aref;
    if (q.l_on[0]) q.l_jar[0] = v[q.l_dof] - q.l_aref[0];
    if (q.l_on[1]) q.l_jar[1] = -v[q.l_dof] - q.l_aref[1];
    if (q.t_on[0] || q.t_on[1]) {
      const wreal jv = wt_tendon_jv(m, lane, v);
      if (q.t_on[0]) q.t_jar[0] = jv - q.t_aref[0];
      if (q.t_on[1]) q.t_jar[1] = -jv - q.t_aref[1];
    }
    if (q.s_on) {
      wreal s = -q.s_aref;
      const int b1 = t.s_body[2 * lane], b2 = t.s_body[2 * lane + 1];
      for (int e = 0; e < 6; e++) s += t.s_a[6 * lane + e] * (t.Vb[6 * b2 + e] - t.Vb[6 * b1 + e]);
      q.s_jar = s;
    }
    if (q.c_on)
      WT_CONE(lane,
        wreal geo[12]; wreal V[6]; wreal x[6];
        const int b1 = c.meta[0]; const int b2 = c.meta[1];
        const wreal* aref = c.X + 6;  // (overwritten by the first Hessian, after the last set_jar)
        for (int e = 0; e < 12; e++) geo[e] = c.geo[e];
        for (int e = 0; e < 6; e++) V[e] = t.Vb[6 * b2 + e] - t.Vb[6 * b1 + e];
        wt_cone_project(x, geo, V, q.c_dim);
        for (int e = 0; e < 6; e++) c.jar[e] = e < q.c_dim ? x[e] - aref[e] : WL(0.0););
    WSYNC();
  };
  set_jar(d.qacc);
  wreal cost = wt_cost(d, t, q, lane, pyr);
  bool factor_valid = false, have_Ma = false, have_jtf = false;
  wreal improvement = 0, Ma_kept = 0, jtf_kept = 0;
  if (have_warm) {  // warm start (mj_fwdConstraint): begin at the previous step's qacc if its cost is lower
    wreal gauss = 0, Mw = 0;
    if (lane < nv) {
      Mw = wt_sym_mulvec_diff<NMAX>(d.M, d.qacc_warm, d.qacc_smooth, nv, lane);
      gauss = WL(0.5) * Mw * (d.qacc_warm[lane] - d.qacc_smooth[lane]);
    }
    gauss = wave_sum(gauss);
    set_jar(d.qacc_warm);
    const wreal cw = gauss + wt_cost(d, t, q, lane, pyr);
    if (cw < cost) {
      cost = cw;
      if (lane < nv) d.qacc[lane] = d.qacc_warm[lane];
      Ma_kept = Mw; have_Ma = true;  // (the first gradient's M (qacc - qacc_smooth))
      WSYNC();
    } else {
      set_jar(d.qacc);
      wt_cost(d, t, q, lane, pyr);  // restore jar / force / zone of the smooth start
    }
  }
  const wreal scale = WL(1.0) / (m.meaninertia * (nv > 1 ? nv : 1));
  int zf_prev = -2, zl0_prev = -2, zl1_prev = -2, zs_prev = -2, zc_prev = -2, zt0_prev = -2, zt1_prev = -2;
  long long tacc = 0;
#define WACC(k) do { if (stamp && lane == 0) { const long long now_ = (long long)__builtin_readcyclecounter(); stamp[k] += now_ - tacc; tacc = now_; } } while (0)
  for (int iter = 0; iter < m.solver_iterations; iter++) {
    if (stamp && lane == 0) tacc = (long long)__builtin_readcyclecounter();
    // gradient = M (qacc - qacc_smooth) - J' force
    wreal g = 0;
    const wreal jtf = wt_jt_force(m, d, t, q, ns, nc, lane);
    jtf_kept = jtf; have_jtf = true;
    if (lane < nv) {
      // (after the first iteration: the product the previous one formed for its cost, at the same qacc)
      const wreal s = have_Ma ? Ma_kept : wt_sym_mulvec_diff<NMAX>(d.M, d.qacc, d.qacc_smooth, nv, lane);
      d.Ma[lane] = s;
      g = s - jtf;
      d.search[lane] = -g;
    }
    const wreal gnorm = sqrt(wave_sum(lane < nv ? g * g : WL(0.0)));
    if (gnorm == 0) break;
    {
      const wreal tol = sizeof(wreal) == 4 ? fmax((wreal)m.solver_tolerance, WL(1e-7)) : (wreal)m.solver_tolerance;
      if (iter > 0 && (scale * improvement < tol || scale * gnorm < tol)) break;
    }
    if (stamp && lane == 0 && iter == 0) stamp[21] = (long long)__builtin_readcyclecounter();
    WACC(32);
    // H depends on the rows' zones only -- and on jar for a cone in its middle (sliding) zone: reuse the factor when nothing moved
    bool refresh;
    {
      const bool moved = (q.f_on && q.f_zone != zf_prev) || (q.l_on[0] && q.l_zone[0] != zl0_prev) || (q.l_on[1] && q.l_zone[1] != zl1_prev) ||
                         (q.s_on && q.s_zone != zs_prev) || (q.c_on && (q.c_zone != zc_prev || q.c_zone == kZoneMiddle)) ||
                         (q.t_on[0] && q.t_zone[0] != zt0_prev) || (q.t_on[1] && q.t_zone[1] != zt1_prev);
      zf_prev = q.f_zone; zl0_prev = q.l_zone[0]; zl1_prev = q.l_zone[1]; zs_prev = q.s_zone; zc_prev = q.c_zone; zt0_prev = q.t_zone[0]; zt1_prev = q.t_zone[1];
      refresh = !factor_valid || __any(moved);
    }
    if (refresh) {
      // X_c = A' Hc A of every cone outside its top zone (one lane per cone)
      if (q.c_on)
        WT_CONE(lane,
          wreal X[21];
          _Pragma("unroll")
          for (int e = 0; e < 21; e++) X[e] = 0;
          if (q.c_zone != kZoneTop) {
            wreal geo[12]; wreal par[12]; wreal x[6]; wreal a[6];
            for (int e = 0; e < 12; e++) { geo[e] = c.geo[e]; par[e] = c.par[e]; }
            for (int e = 0; e < 6; e++) x[e] = c.jar[e];
            const int dim = q.c_dim;
            if (pyr) {
              // Hc = D sum over the active edges of e e', e = a_0 +- f_j a_j
              wreal a0[6];
              wt_cone_gen(a0, geo, 0);
              const int mask = q.c_zone >> 2;
              _Pragma("unroll")
              for (int j = 1; j < 6; j++)
                if (j < dim) {
                  wt_cone_gen(a, geo, j);
                  wreal e[6];
                  if (mask & (1 << (2 * (j - 1)))) { for (int k = 0; k < 6; k++) e[k] = a0[k] + par[j] * a[k]; sym6_rank1(X, par[6], e); }
                  if (mask & (2 << (2 * (j - 1)))) { for (int k = 0; k < 6; k++) e[k] = a0[k] - par[j] * a[k]; sym6_rank1(X, par[6], e); }
                }
              (void)x;
            } else if (q.c_zone == kZoneBottom) {
              _Pragma("unroll")
              for (int j = 0; j < 6; j++)
                if (j < dim) { wt_cone_gen(a, geo, j); if (j < 3) sym6_rank1(X, par[6 + j], a); else sym3_rank1(X, par[6 + j], a); }
            } else {
              // Hc = Dm w w' - c (S^2 - y y'),  w = (mu, -mu f_j u_j), y = (0, f_j u_j), u = U_t / |U_t|, c = Dm NT mu / T
              const wreal mu = par[0];
              wreal U[6]; wreal T = 0;
              U[0] = x[0] * mu;
              _Pragma("unroll")
              for (int j = 1; j < 6; j++) { U[j] = j < dim ? x[j] * par[j] : WL(0.0); T += U[j] * U[j]; }
              T = sqrt(T);
              const wreal iT = WL(1.0) / T;
              const wreal Dm = q.c_Dm; const wreal NT = U[0] - mu * T; const wreal cc = Dm * NT * mu * iT;
              wreal p[6] = {0, 0, 0, 0, 0, 0}; wreal r[6] = {0, 0, 0, 0, 0, 0};
              _Pragma("unroll")
              for (int j = 0; j < 6; j++)
                if (j < dim) {
                  wt_cone_gen(a, geo, j);
                  const wreal yj = j == 0 ? WL(0.0) : par[j] * U[j] * iT;
                  const wreal wj = j == 0 ? mu : -mu * yj;
                  for (int e = 0; e < 6; e++) { p[e] += wj * a[e]; r[e] += yj * a[e]; }
                  if (j >= 1) { if (j < 3) sym6_rank1(X, -cc * par[j] * par[j], a); else sym3_rank1(X, -cc * par[j] * par[j], a); }
                }
              sym6_rank1(X, Dm, p);
              sym6_rank1(X, cc, r);
            }
          }
          for (int e = 0; e < 21; e++) c.X[e] = X[e];);
      for (int e = lane; e < nv * (nv + 1) / 2; e += 64) d.H[e] = d.M[e];
      WSYNC();
      WACC(33);
      // row i of H, one lane per dof: z = sum over the contacts below dof i of X_c cdof_i, H_ij = M_ij + cdof_j . z on the chain
      {
        const wreal lim = (q.l_on[0] && q.l_zone[0] == kZoneBottom ? q.l_D[0] : WL(0.0)) + (q.l_on[1] && q.l_zone[1] == kZoneBottom ? q.l_D[1] : WL(0.0));
        const wreal dlim = wt_pull_from_joint(lim, q.jnt_of_dof);
        if (lane < nv) {
          wreal cd[6], z[6] = {0, 0, 0, 0, 0, 0};
          for (int e = 0; e < 6; e++) cd[e] = d.cdof[6 * lane + e];
          // contacts against static geoms: every dof j <= i on the chain of i is on the contact's chain too, so their X_c cdof_i
          // add up before the chain is walked once; a contact between two moving bodies walks its own dof set
          bool any = false;
          const unsigned lowmask = (2u << lane) - 1u;  // dofs j <= i
          for (int i = 0; i < ns; i++) {
            const unsigned mp = (unsigned)__builtin_amdgcn_readlane((int)q.s_mp, i), mm = (unsigned)__builtin_amdgcn_readlane((int)q.s_mm, i);
            const int sg = (int)((mp >> lane) & 1u) - (int)((mm >> lane) & 1u);
            const wreal D = t.s_fd[2 * i + 1];
            const bool pair = __builtin_amdgcn_readlane(q.s_pair, i) != 0;
            if (D != 0) {  // (wave-uniform)
              wreal a[6], ja = 0;
              for (int e = 0; e < 6; e++) { a[e] = t.s_a[6 * i + e]; ja += a[e] * cd[e]; }
              if (mm == 0 && !pair) { for (int e = 0; e < 6; e++) z[e] += (wreal)sg * D * ja * a[e]; any |= sg != 0; }
              else if (sg != 0) { wreal zc[6]; for (int e = 0; e < 6; e++) zc[e] = D * ja * a[e]; wt_pair_rows(d, lane, zc, mp, mm, sg, lowmask); }
            }
          }
          for (int i = 0; i < nc; i++) {
            const unsigned mp = (unsigned)__builtin_amdgcn_readlane((int)q.c_mp, i), mm = (unsigned)__builtin_amdgcn_readlane((int)q.c_mm, i);
            const int sg = (int)((mp >> lane) & 1u) - (int)((mm >> lane) & 1u);
            const int zone_i = __builtin_amdgcn_readlane(q.c_zone, i);
            const bool pair = __builtin_amdgcn_readlane(q.c_pair, i) != 0;
            if (zone_i == kZoneTop) continue;  // X_c = 0 (wave-uniform)
            WT_CONE(i,
              if (mm == 0 && !pair) { if (sg != 0) { sym6_mulvec_acc(z, c.X, cd); any = true; } }
              else if (sg != 0) { wreal zc[6] = {0, 0, 0, 0, 0, 0}; sym6_mulvec_acc(zc, c.X, cd); wt_pair_rows(d, lane, zc, mp, mm, sg, lowmask); });
          }
          const wreal diag = (q.f_on && q.f_zone == kZoneBottom ? q.f_D : WL(0.0)) + dlim;
          if (any) {
            unsigned chain = m.body_dofmask[m.dof_bodyid[lane]] & lowmask;  // dofs j <= i on the chain of i
            while (chain) {
              const int j = __ffs((int)chain) - 1;
              chain &= chain - 1;
              wreal s = 0;
              for (int e = 0; e < 6; e++) s += d.cdof[6 * j + e] * z[e];
              d.H[wt_tri(lane, j)] += s;
            }
          }
          if (diag != 0) d.H[wt_tri(lane, lane)] += diag;
        }
        // limits of fixed tendons: H += D coef coef' over the tendon's dofs (row = this lane's dof, columns j <= row)
        for (int i = 0; i < m.ntendon; i++) {
          const wreal Di = (wreal)__shfl((q.t_on[0] && q.t_zone[0] == kZoneBottom ? q.t_D[0] : WL(0.0)) + (q.t_on[1] && q.t_zone[1] == kZoneBottom ? q.t_D[1] : WL(0.0)), i, 64);
          if (Di == 0) continue;  // (wave-uniform)
          const unsigned tm = m.tendon_dofmask[i];
          if (lane < nv && ((tm >> lane) & 1u)) {
            const wreal ci = wt_tendon_coef(m, i, lane);
            unsigned cols = tm & ((2u << lane) - 1u);
            while (cols) {
              const int j = __ffs((int)cols) - 1;
              cols &= cols - 1;
              d.H[wt_tri(lane, j)] += Di * ci * wt_tendon_coef(m, i, j);
            }
          }
        }
      }
      WSYNC();
      if (stamp && lane == 0 && iter == 0) stamp[22] = (long long)__builtin_readcyclecounter();
      WACC(34);
      if (!wt_chol<NMAX>((const wlds_f64*)d.H, (wlds_f64*)d.H, (wlds_f64*)d.dinv, nv, lane)) { if (lane == 0) d.counters[2] |= 16; WSYNC(); break; }
      factor_valid = true;
    }
    wt_chol_solve<NMAX>((wlds_f64*)d.search, (const wlds_f64*)d.H, (const wlds_f64*)d.dinv, nv, lane);
    if (stamp && lane == 0 && iter == 0) stamp[23] = (long long)__builtin_readcyclecounter();
    WACC(35);
    // jv = J search (registers); Gauss part along the ray
    wt_body_vectors(m, d, t, d.search, lane);
    wreal f_jv = 0, l_jv[2] = {0, 0}, s_jv = 0, c_jv[6] = {0, 0, 0, 0, 0, 0}, c_x0[6] = {0, 0, 0, 0, 0, 0}, c_par[12];
#pragma unroll
    for (int e = 0; e < 12; e++) c_par[e] = 0;
    if (q.f_on) f_jv = d.search[lane];
    if (q.l_on[0]) l_jv[0] = d.search[q.l_dof];
    if (q.l_on[1]) l_jv[1] = -d.search[q.l_dof];
    wreal t_jv = 0;
    if (q.t_on[0] || q.t_on[1]) t_jv = wt_tendon_jv(m, lane, d.search);
    if (q.s_on) {
      const int b1 = t.s_body[2 * lane], b2 = t.s_body[2 * lane + 1];
      for (int e = 0; e < 6; e++) s_jv += t.s_a[6 * lane + e] * (t.Vb[6 * b2 + e] - t.Vb[6 * b1 + e]);
    }
    if (q.c_on)
      WT_CONE(lane,
        wreal geo[12]; wreal V[6];
        const int b1 = c.meta[0]; const int b2 = c.meta[1];
        for (int e = 0; e < 12; e++) { geo[e] = c.geo[e]; c_par[e] = c.par[e]; }
        for (int e = 0; e < 6; e++) { V[e] = t.Vb[6 * b2 + e] - t.Vb[6 * b1 + e]; c_x0[e] = c.jar[e]; }
        wt_cone_project(c_jv, geo, V, q.c_dim););
    wreal q1 = 0, q2 = 0;
    if (lane < nv) {
      const wreal s = wt_sym_mulvec<NMAX>(d.M, d.search, nv, lane);
      q1 = d.search[lane] * d.Ma[lane];
      q2 = d.search[lane] * s;
    }
    q1 = wave_sum(q1); q2 = wave_sum(q2);
    WACC(36);
    // exact line search: safeguarded 1-D Newton on the (convex, piecewise quadratic) restriction; all rows from registers
    auto line = [&](wreal alpha, wreal& g1, wreal& h2) {
      g1 = 0; h2 = 0;
      if (q.f_on) {
        const wreal v = f_jv, x = q.f_jar + alpha * v, f = q.f_fl, R = q.f_R, D = q.f_D;
        if (x <= -R * f) g1 += -f * v;
        else if (x >= R * f) g1 += f * v;
        else { g1 += D * x * v; h2 += D * v * v; }
      }
#pragma unroll
      for (int s = 0; s < 2; s++)
        if (q.l_on[s]) {
          const wreal v = l_jv[s], x = q.l_jar[s] + alpha * v, D = q.l_D[s];
          if (x < 0) { g1 += D * x * v; h2 += D * v * v; }
        }
#pragma unroll
      for (int s = 0; s < 2; s++)
        if (q.t_on[s]) {
          const wreal v = s == 0 ? t_jv : -t_jv, x = q.t_jar[s] + alpha * v, D = q.t_D[s];
          if (x < 0) { g1 += D * x * v; h2 += D * v * v; }
        }
      if (q.s_on) {
        const wreal v = s_jv, x = q.s_jar + alpha * v, D = q.s_D;
        if (x < 0) { g1 += D * x * v; h2 += D * v * v; }
      }
      if (q.c_on) {
        wreal cg, ch;
        if (pyr) wt_pyr_line(c_x0, c_jv, alpha, c_par, q.c_dim, cg, ch);
        else wt_cone_line(c_x0, c_jv, alpha, c_par, q.c_Dm, q.c_dim, cg, ch);
        g1 += cg; h2 += ch;
      }
    };
    wreal lo = 0, hi = -1, alpha = 0, d1, d2, g0, h0;
    line(WL(0.0), g0, h0);
    d1 = wave_sum(g0) + q1; d2 = wave_sum(h0) + q2;
    const wreal d10 = fabs(d1);
    wreal gtol = m.solver_tolerance * kLsTolerance * sqrt(wave_sum(lane < nv ? d.search[lane] * d.search[lane] : WL(0.0))) / scale;
    if (sizeof(wreal) == 4) gtol = fmax(gtol, WL(1e-4) * d10);
    wreal step1 = WL(1e30), step2 = WL(1e30);  // the last step and the one before (rtsafe safeguard, oracle/contact.inc)
    for (int ls = 0; ls < 50 && d10 >= gtol; ls++) {
      wreal an = alpha - d1 / d2;
      if (!(an > lo) || (hi >= 0 && !(an < hi))) an = hi >= 0 ? WL(0.5) * (lo + hi) : 2 * alpha + 1;
      else if (hi >= 0 && fabs(an - alpha) > WL(0.5) * step2) an = WL(0.5) * (lo + hi);
      if (an == alpha) break;
      step2 = step1; step1 = fabs(an - alpha);
      alpha = an;
      line(alpha, g0, h0);
      d1 = wave_sum(g0) + q1 + alpha * q2; d2 = wave_sum(h0) + q2;
      if (fabs(d1) < gtol) break;
      if (d1 < 0) lo = alpha; else hi = alpha;
      if (stamp && lane == 0) stamp[39]++;
    }
    WACC(37);
    if (stamp && lane == 0 && iter == 0) stamp[24] = (long long)__builtin_readcyclecounter();
    if (lane < nv) d.qacc[lane] += alpha * d.search[lane];
    if (q.f_on) q.f_jar += alpha * f_jv;
    if (q.l_on[0]) q.l_jar[0] += alpha * l_jv[0];
    if (q.l_on[1]) q.l_jar[1] += alpha * l_jv[1];
    if (q.t_on[0]) q.t_jar[0] += alpha * t_jv;
    if (q.t_on[1]) q.t_jar[1] -= alpha * t_jv;
    if (q.s_on) q.s_jar += alpha * s_jv;
    if (q.c_on) WT_CONE(lane, for (int e = 0; e < 6; e++) c.jar[e] = c_x0[e] + alpha * c_jv[e];);
    WSYNC();
    wreal gauss = 0;
    if (lane < nv) {
      const wreal s = wt_sym_mulvec_diff<NMAX>(d.M, d.qacc, d.qacc_smooth, nv, lane);
      gauss = WL(0.5) * s * (d.qacc[lane] - d.qacc_smooth[lane]);
      Ma_kept = s;
    }
    have_Ma = true;
    gauss = wave_sum(gauss);
    const wreal newcost = gauss + wt_cost(d, t, q, lane, pyr);
    have_jtf = false;  // (the forces moved)
    improvement = cost - newcost;
    cost = newcost;
    WACC(38);
    if (stamp && lane == 0) stamp[20] = iter + 1;
  }
#undef WACC
  // (a loop left at its convergence test has J' force of the final forces in hand)
  const wreal jtf = have_jtf ? jtf_kept : wt_jt_force(m, d, t, q, ns, nc, lane);
  if (lane < nv) d.qfrc_constraint[lane] = jtf;
  WSYNC();
}


// ---- mj_forward up to the constraint solve on the tree path (wf_forward with the Jacobian-free constraint stages)
template <int NMAX, class MODEL, class TASK>
__device__ __forceinline__ void wt_forward(const MODEL& m, const TASK& tk, WaveData& d, TreeData& t, int lane, bool& bad_ctrl,
                                           long long* stamp, bool have_warm) {
  const int nv = m.nv;
  // Order: the smooth dynamics run BEFORE collision and the rows (they do not depend on them): the RNE work arrays, cinert
  // and cdof_dot are dead when the contact lists are born, so the lists live in the same LDS (wave_carve_tree)
  WSTAMP(1);
  wf_kinematics(m, tk, d, lane);
  WSYNC();
  WSTAMP(2);
  wf_compos(m, d, lane);
  WSTAMP(3);
  wt_crb(m, d, lane);
  WSTAMP(4);
  if (!wt_chol<NMAX>((const wlds_f64*)d.M, (wlds_f64*)d.L, (wlds_f64*)d.Ldinv, nv, lane)) { if (lane == 0) d.counters[2] |= 16; }
  WSTAMP(5);
  wf_comvel(m, d, lane);
  WSTAMP(6);
  wf_smooth_forces(m, d, lane, bad_ctrl);
  WSTAMP(7);
  wt_chol_solve<NMAX>((wlds_f64*)d.qacc_smooth, (const wlds_f64*)d.L, (const wlds_f64*)d.Ldinv, nv, lane);
  WSTAMP(8);
  wt_collision(m, d, t, lane);
  WSTAMP(9);
  TreeRows q;
  wt_make_constraint(m, d, t, q, lane);
  WSTAMP(10);
  wt_constraint_newton<NMAX>(m, d, t, q, lane, stamp, have_warm);
  WSTAMP(11);
}

// LDS footprint / layout of one candidate on the tree path. Arrays with disjoint lifetimes share storage:
//   U1: ximat | xanchor | xaxis (kinematics -> compos)   over   cvel | cdof_dot (comvel -> residual / RNE); Vb (Newton) over cdof_dot
//   R : cinert | cacc | cfrc | cfrc_sub, crb over cacc|cfrc (compos -> RNE)   over   the contact lists (collision -> Newton)
//       over   foot_xpos | residual | terms | scal | the cost norms' per-entry scratch (sensor stage)
//   qfrc_passive / bias / actuator (smooth forces) over search / Ma / tmpv (Newton, Euler)
// M and the Cholesky factors are packed lower triangles. P (node times) is the only run-time size: it comes last.
__host__ __device__ inline size_t wave_tree_u1(int nv, int nbody, int njnt) {
  const size_t a = 6 * (size_t)nbody + 6 * (size_t)nv, b = 9 * (size_t)nbody + 6 * (size_t)njnt;
  return a > b ? a : b;
}
__host__ __device__ inline size_t wave_tree_r(int nbody, int nr, int nterm, int caps, int capc) {
  size_t r = 28 * (size_t)nbody;
  if (tree_lds_elems(caps, capc) > r) r = tree_lds_elems(caps, capc);
  const size_t sensor = 12 + 2 * (size_t)nr + nterm + 8;
  return sensor > r ? sensor : r;
}
__host__ __device__ inline size_t wave_lds_elems_tree(int nq, int nv, int nu, int nbody, int njnt, int nsite, int nr, int nterm, int P, bool xfrc = false,
                                                      int caps = kTreeMaxSimple, int capc = kTreeMaxCone) {
  size_t n = 0;
  n += nq + nv + nu + nu + nv;                                       // qpos qvel ctrl actuator_force qacc_warm
  n += 3 * nbody + 4 * nbody + 9 * nbody + 3 * nbody + 3 * nsite + 3 * nbody + 3;  // xpos xquat xmat xipos site_xpos subtree_com subtree_linvel
  n += 6 * nv;                                                       // cdof
  n += (size_t)nv * (nv + 1) + nv;                                   // M, H (packed) + reciprocal pivots
  n += wave_tree_u1(nv, nbody, njnt);
  n += wave_tree_r(nbody, nr, nterm, caps, capc);
  n += 7 * nv;                                                       // qfrc_smooth qacc_smooth qacc qfrc_constraint search Ma tmpv
  n += 4 * sizeof(int) / sizeof(wreal) + 1;                          // counters
  n += xfrc ? 6 * nbody : 0;
  n += P;
  return n + 4;
}
template <class MODEL, class TASK>
__device__ __forceinline__ WaveData wave_carve_tree(unsigned char* smem_raw, const MODEL& m, const TASK& tk, int P, wreal*& ltimes, bool xfrc, TreeData& t,
                                                    int caps = kTreeMaxSimple, int capc = kTreeMaxCone) {
  const int nq = m.nq, nv = m.nv, nu = m.nu, nb = m.nbody, nj = m.njnt, ns = m.nsite, nr = tk.nr;
  wreal* p = reinterpret_cast<wreal*>(smem_raw);
  auto take = [&](size_t n) { wreal* q = p; p += n; return q; };
  WaveData d;
  d.qpos = take(nq); d.qvel = take(nv); d.ctrl = take(nu); d.actuator_force = take(nu); d.qacc_warm = take(nv);
  d.xpos = take(3 * nb); d.xquat = take(4 * nb); d.xmat = take(9 * nb); d.xipos = take(3 * nb); d.site_xpos = take(3 * ns);
  d.subtree_com = take(3 * nb); d.subtree_linvel = take(3);
  d.cdof = take(6 * nv);
  d.M = take((size_t)nv * (nv + 1) / 2); d.H = take((size_t)nv * (nv + 1) / 2); d.Ldinv = take(nv); d.dinv = d.Ldinv; d.L = d.H;
  {
    wreal* u1 = take(wave_tree_u1(nv, nb, nj));
    d.cvel = u1; d.cdof_dot = u1 + 6 * nb;
    d.ximat = u1; d.xanchor = u1 + 9 * nb; d.xaxis = d.xanchor + 3 * nj;
    t.Vb = d.cdof_dot;  // 6 nbody <= 6 nv for a model the tree path takes? not in general: checked on the host (wave_tree_fits)
  }
  {
    wreal* r = take(wave_tree_r(nb, nr, tk.nterm, caps, capc));
    d.cinert = r; d.cacc = r + 10 * nb; d.cfrc = d.cacc + 6 * nb; d.cfrc_sub = d.cfrc + 6 * nb; d.crb = d.cacc;
    wreal* vb = t.Vb;
    t = tree_carve(r, caps, capc);
    t.Vb = vb;
    d.foot_xpos = r; d.residual = r + 12; d.terms = d.residual + nr; d.scal = d.terms + tk.nterm; d.efc_J = d.scal + 8;
  }
  d.qfrc_smooth = take(nv); d.qacc_smooth = take(nv); d.qacc = take(nv); d.qfrc_constraint = take(nv);
  d.search = take(nv); d.Ma = take(nv); d.tmpv = take(nv);
  d.qfrc_passive = d.search; d.qfrc_bias = d.Ma; d.qfrc_actuator = d.tmpv;
  d.grad = nullptr; d.Ms = nullptr;
  d.efc_D = d.efc_R = d.efc_aref = d.efc_floss = d.efc_force = d.jar = d.jv = d.efc_pos = d.efc_margin = nullptr;
  d.efc_type = d.efc_id = d.efc_zone = nullptr;
  d.coneH = nullptr; d.con = nullptr;
  d.counters = reinterpret_cast<int*>(take(4 * sizeof(int) / sizeof(wreal) + 1));
  d.xfrc = xfrc ? take(6 * nb) : nullptr;
  ltimes = take(P);
  return d;
}

} }  // namespace mjpcx::WAVE_NS
