// tree_kernel.h -- rollout kernel of a REGISTERED model (per-precision include, see rollout_wave.h): the step function of
// rollout_wave_kernel<NMAX, TREE = true> (wave_rollout_body), with
//   * the model's hot arrays staged ONCE per workgroup into LDS behind compile-time offsets (lds_model.h): no global
//     load, no pointer held in SGPRs and no run-time size in the step loop;
//   * up to 8 wavefronts per workgroup sharing that image, each with its own candidate arena;
//   * persistent wavefronts drawing candidates from an atomic counter: the image is staged once per CU, and a long rollout (more
//     Newton iterations) neither holds a whole workgroup's LDS back nor decides when the launch ends.
namespace mjpcx { namespace WAVE_NS {

// mode bit 5 (32): roll out only the candidates whose failure[] carries kQFallback -- the ones rollout_quad_kernel (quad_kernel.h) handed on
// mode bit 4 (16): dynamic candidate hand-out (the default; 0 = static grid stride, for A/B runs: tools/ab_mode.sh)
// mode bit 3 (8): poison every arena before each rollout (uninitialised-read detector, tests/test_gpu_quadruped.py)
// mode bit 1: self-check -- compare the staged image with the generic model's arrays, mismatches are counted in work[1]
//             and the launch rolls nothing out (tuning / bring-up aid, MJPCX_TREE_CHECK=1)
// BIG: second pass -- only the candidates the first pass flagged "contact list full" (failure bits, mjpcx.h), with the large lists
#ifndef TREE_KERNEL_THREADS
#define TREE_KERNEL_THREADS 512  // 8 wavefronts of 256 registers
#endif
template <class C, bool BIG = false>
__global__ __launch_bounds__(TREE_KERNEL_THREADS) void rollout_tree_kernel(const WModel m_in, const WTask tk_in, const RolloutArgs<wreal> a,
                                                           const unsigned char* __restrict__ image, unsigned blob_bytes, unsigned arena_bytes,
                                                           int* work, int mode, wreal* __restrict__ cone_slabs) {
  typedef LdsLayout<C, wreal> L;
  const int tid = threadIdx.x, nth = blockDim.x;
  {
    const uint4* src = reinterpret_cast<const uint4*>(image);
    uint4* dst = reinterpret_cast<uint4*>(mjpcx_lds);
    for (unsigned i = tid; i < L::kBytes / 16; i += nth) dst[i] = src[i];
    const uint4* bs = reinterpret_cast<const uint4*>(tk_in.blob);
    uint4* bd = reinterpret_cast<uint4*>(mjpcx_lds + L::kBytes);
    for (unsigned i = tid; i < blob_bytes / 16; i += nth) bd[i] = bs[i];
  }
  __syncthreads();
  const LdsModelT<C, wreal> m(m_in);
  const LdsTaskT<C, wreal> tk(tk_in, reinterpret_cast<const wreal*>(mjpcx_lds + L::kBytes));
  const int wave = tid >> 6, lane = tid & 63;
  unsigned char* arena = mjpcx_lds + L::kBytes + blob_bytes + (unsigned)wave * arena_bytes;
  if (mode & 2) {
    int bad = 0;
#define CHK(name, n) for (int i = tid; i < (n); i += nth) bad += m.name[i] != m_in.name[i];
    CHK(body_parentid, C::NB) CHK(body_rootid, C::NB) CHK(body_jntnum, C::NB) CHK(body_jntadr, C::NB) CHK(body_dofnum, C::NB) CHK(body_dofadr, C::NB)
    CHK(body_mocapid, C::NB) CHK(body_pos, 3 * C::NB) CHK(body_quat, 4 * C::NB) CHK(body_ipos, 3 * C::NB) CHK(body_iquat, 4 * C::NB) CHK(body_mass, C::NB)
    CHK(body_inertia, 3 * C::NB) CHK(body_invweight0, 2 * C::NB) CHK(body_subtreemass, C::NB)
    CHK(jnt_type, C::NJ) CHK(jnt_qposadr, C::NJ) CHK(jnt_dofadr, C::NJ) CHK(jnt_bodyid, C::NJ) CHK(jnt_limited, C::NJ) CHK(jnt_pos, 3 * C::NJ)
    CHK(jnt_axis, 3 * C::NJ) CHK(jnt_stiffness, C::NJ) CHK(jnt_range, 2 * C::NJ) CHK(jnt_margin, C::NJ) CHK(jnt_solref, 2 * C::NJ) CHK(jnt_solimp, 5 * C::NJ)
    CHK(dof_bodyid, C::NV) CHK(dof_jntid, C::NV) CHK(dof_parentid, C::NV) CHK(dof_armature, C::NV) CHK(dof_damping, C::NV) CHK(dof_frictionloss, C::NV)
    CHK(dof_invweight0, C::NV) CHK(dof_solref, 2 * C::NV) CHK(dof_solimp, 5 * C::NV) CHK(qpos0, C::NQ) CHK(qpos_spring, C::NQ)
    CHK(site_bodyid, C::NS) CHK(site_pos, 3 * C::NS) CHK(actuator_trnid, C::NU) CHK(actuator_biastype, C::NU) CHK(actuator_ctrllimited, C::NU)
    CHK(actuator_forcelimited, C::NU) CHK(actuator_gear, C::NU) CHK(actuator_gainprm, 3 * C::NU) CHK(actuator_biasprm, 3 * C::NU)
    CHK(actuator_ctrlrange, 2 * C::NU) CHK(actuator_forcerange, 2 * C::NU) CHK(geom_type, C::NG) CHK(geom_bodyid, C::NG) CHK(geom_contype, C::NG)
    CHK(geom_conaffinity, C::NG) CHK(geom_condim, C::NG) CHK(geom_priority, C::NG) CHK(geom_size, 3 * C::NG) CHK(geom_pos, 3 * C::NG)
    CHK(geom_quat, 4 * C::NG) CHK(geom_margin, C::NG) CHK(key_qpos, C::NKEY * C::NQ) CHK(body_subtree_mask, C::NB) CHK(body_dofmask, C::NB)
    CHK(level_body, m_in.level_start[m_in.nlevel]) CHK(static_geom, C::NSG) CHK(dynamic_geom, C::NDG) CHK(ray_geom, C::NRAY)
#undef CHK
    for (int i = tid; i < C::NTERM; i += nth) bad += tk.dim_norm_residual[i] != tk_in.dim_norm_residual[i] || tk.norm[i] != tk_in.norm[i] || tk.term_off[i] != tk_in.term_off[i];
    for (int i = tid; i < C::NR; i += nth) bad += tk.res_term[i] != tk_in.res_term[i];
    for (int i = tid; i < C::NTRACE; i += nth) bad += tk.trace_site[i] != tk_in.trace_site[i];
    for (int i = tid; i < (int)(blob_bytes / sizeof(wreal)); i += nth) bad += !(tk.blob[i] == tk_in.blob[i]) && tk.blob[i] == tk.blob[i];
    if (bad) atomicAdd(work + 1, bad);
    return;
  }
  // Persistent wavefronts. Every wavefront starts on candidate (workgroup, wavefront); after that, mode bit 4 (16) hands the
  // remaining candidates out through one atomic counter (work[2], zeroed by the host before the launch) -- a wavefront that drew
  // cheap rollouts (few Newton iterations) takes more of them, so the launch ends with the mean, not with the unluckiest
  // stride; without the bit, the static grid stride. Results do not depend on which wavefront rolls a candidate out.
  // cones beyond the LDS list go to this wavefront's slab in global memory (wave_tree.h)
  const int total_waves = gridDim.x * (nth >> 6);
  wreal* slab = cone_slabs ? cone_slabs + (size_t)(blockIdx.x * (nth >> 6) + wave) * (size_t)((kTreeMaxConeTotal - kTreeMaxCone) * kConeRec) : nullptr;
  const bool dynamic = !BIG && (mode & 16);
  int cand = blockIdx.x * (nth >> 6) + wave;
  while (cand < a.N) {
    if (mode & 8) {  // bring-up aid (MJPCX_TREE_MODE=8): poison the arena -- a read of storage this rollout never wrote shows up as NaN / -1
      for (unsigned i = lane; i < arena_bytes / 4; i += 64) reinterpret_cast<unsigned*>(arena)[i] = 0xFFFFFFFFu;
      WSYNC();
    }
    if constexpr (BIG) {
      if (a.failure[cand] & (32 << 8))  // wave-uniform
        wave_rollout_body<C::NMAX, true, kTreeMaxSimpleBig, kTreeMaxConeBig>(m, tk, a, arena, cand, lane);
    } else {
      if (!(mode & 32) || (a.failure[cand] & kQFallback))  // wave-uniform
        wave_rollout_body<C::NMAX, true>(m, tk, a, arena, cand, lane, slab);
    }
    if (dynamic) {
      int next = 0;
      if (lane == 0) next = atomicAdd(work + 2, 1);
      cand = total_waves + __builtin_amdgcn_readfirstlane(next);
    } else {
      cand += total_waves;
    }
  }
}

} }  // namespace mjpcx::WAVE_NS
