// tds_chain_model.h — the constant table of the serial-chain kernel (tds_chain.hip) and the host-side detection of the models it
// takes.  Included by tds_device_model.h behind the definition of DevModel (like tds_oct_model.h).
#pragma once

// The constants of a fixed-base serial chain (DevModel::chain) as the kernel of tds_chain.hip reads them: one flat table of
// compute scalars in DevModel::oct_tab (a model is a star or a chain, never both), copied into LDS at the top of a launch.
// Offsets in scalars.  Per link a record of LSTR scalars (a stride whose eight records start on different LDS banks), then the
// model's scalars.  Links beyond the chain's length are massless, motionless identity links: they add nothing to the sums
// that run over the lanes of an environment.
struct TdsChainTab {
  static constexpr int LSTR = 58;
  // the link's record.  XT: X_T of the link, for link 0 already composed with base_X_world (kinematics.hpp:92)
  static constexpr int S = 0, XT = 6 /* rotation 9 | translation 3 */, MASS = 18, COM = 19, INER = 22, STIFF = 31, DAMP = 32,
                       // the joint rotation as Rodrigues' formula about the unit axis n (see TdsOctTab): n (3) | n n^T (6) | 1 revolute, 0 prismatic
                       NAX = 33, NN = 36, ROTF = 42, VIS = 43 /* X_visual: rotation 9 | translation 3 */;
  static constexpr int SC = 8 * LSTR;
  static constexpr int DT = 0, GRAV = 1 /* base_X_world.rot * gravity */, BASE_R8 = 4, NUM_VISUALS = 5, PACK_VISUALS = 6, OUTPUT_DIM = 7, NUM_LINKS = 8,
                       XT_IDENT = 9,  // 1: the X_T rotation of every link (link 0: with the base) is the identity
                       VIS_IDENT = 10;  // 1: every visual's rotation is the identity (no R X_visual product)
  static constexpr int TOTAL = SC + 12;
};

static_assert(TdsChainTab::TOTAL <= TDS_OCT_TAB_CAP, "DevModel::oct_tab is too small for the chain table");

// sets d->chain (and fills d->oct_tab) where the model is a chain the kernel of tds_chain.hip is built for: fixed base, 2 .. 8
// links, link i the child of link i - 1 and the owner of coordinate / velocity / torque i, 1-dof joints, joint torques given
// directly (TDS_STEP_TAU), no plane and no contact geometry in play, one body, no reward rule; visual v on link v.  BASELINE
// configs 1 and 2 (cartpole, pendulum5) are such chains; create-time option chain = 0 keeps them on the general kernel.
template <typename T>
static void tds_chain_detect(const tds_model_t *m, DevModel<T> *d) {
  d->chain = 0;
  const int n = m->num_links;
  if (sizeof(T) != 8 || tds_opt_now(TDS_OPT_CHAIN) == 0 || d->oct || d->quad) return;
  if (m->step_mode != TDS_STEP_TAU || m->has_plane || m->is_floating || m->num_bodies >= 2 || n < 2 || n > 8 || m->dof_q != n ||
      m->dof_qd != n || m->action_dim != n || m->input_dim != 3 * n || m->reward_mode != TDS_REWARD_NONE)
    return;
  if (d->num_links != n) return;  // (fixed links folded away, pseudo links: not this kernel's models)
  for (int i = 0; i < n; ++i) {
    const tds_link_t &l = m->links[i];
    if (l.parent != i - 1 || l.q_index != i || l.qd_index != i || l.joint_type < TDS_JOINT_PRISMATIC_X ||
        l.joint_type > TDS_JOINT_REVOLUTE_AXIS)
      return;
  }
  if (m->pack_visuals && m->num_visuals > n) return;
  if (m->pack_visuals)
    for (int v = 0; v < m->num_visuals; ++v)
      if (m->visuals[v].link != v) return;
  const int want = 2 * n + (m->pack_visuals ? 7 * m->num_visuals + 1 : 0);
  if (m->output_dim < want) return;
  using TB = TdsChainTab;
  T *const t = d->oct_tab;
  for (int i = 0; i < TB::TOTAL; ++i) t[i] = T(0);
  bool ident = true;
  for (int li = 0; li < 8; ++li) {
    T *const r = t + li * TB::LSTR;
    double XR[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1}, Xt[3] = {0, 0, 0};
    if (li < n) {
      const tds_link_t &l = m->links[li];
      for (int k = 0; k < 6; ++k) r[TB::S + k] = (T)l.S[k];
      if (li == 0) {  // base_X_world * X_T (kinematics.hpp:92)
        for (int a = 0; a < 3; ++a) {
          for (int b = 0; b < 3; ++b) {
            double acc = 0.0;
            for (int c = 0; c < 3; ++c) acc += m->base_X_world_rot[3 * a + c] * l.X_T_rot[3 * c + b];
            XR[3 * a + b] = acc;
          }
          double acc = m->base_X_world_trans[a];
          for (int c = 0; c < 3; ++c) acc += m->base_X_world_rot[3 * a + c] * l.X_T_trans[c];
          Xt[a] = acc;
        }
      } else {
        for (int k = 0; k < 9; ++k) XR[k] = l.X_T_rot[k];
        for (int k = 0; k < 3; ++k) Xt[k] = l.X_T_trans[k];
      }
      r[TB::MASS] = (T)l.mass;
      for (int k = 0; k < 3; ++k) r[TB::COM + k] = (T)l.com[k];
      for (int k = 0; k < 9; ++k) r[TB::INER + k] = (T)l.inertia[k];
      r[TB::STIFF] = (T)l.stiffness;
      r[TB::DAMP] = (T)l.damping;
      const double ax[3] = {l.S[0], l.S[1], l.S[2]};
      const double ax2 = ax[0] * ax[0] + ax[1] * ax[1] + ax[2] * ax[2];
      const bool revolute = l.joint_type >= TDS_JOINT_REVOLUTE_X && l.joint_type <= TDS_JOINT_REVOLUTE_AXIS && ax2 > 0.0;
      const double inv = ax2 > 0.0 ? 1.0 / sqrt(ax2) : 0.0;
      const double nn[3] = {revolute ? ax[0] * inv : 0.0, revolute ? ax[1] * inv : 0.0, revolute ? ax[2] * inv : 0.0};
      for (int k = 0; k < 3; ++k) r[TB::NAX + k] = (T)nn[k];
      r[TB::NN + 0] = (T)(nn[0] * nn[0]);
      r[TB::NN + 1] = (T)(nn[0] * nn[1]);
      r[TB::NN + 2] = (T)(nn[0] * nn[2]);
      r[TB::NN + 3] = (T)(nn[1] * nn[1]);
      r[TB::NN + 4] = (T)(nn[1] * nn[2]);
      r[TB::NN + 5] = (T)(nn[2] * nn[2]);
      r[TB::ROTF] = revolute ? T(1) : T(0);
      if (m->pack_visuals && li < m->num_visuals) {
        for (int k = 0; k < 9; ++k) r[TB::VIS + k] = (T)m->visuals[li].X_rot[k];
        for (int k = 0; k < 3; ++k) r[TB::VIS + 9 + k] = (T)m->visuals[li].X_trans[k];
      }
    }
    for (int k = 0; k < 9; ++k) {
      r[TB::XT + k] = (T)XR[k];
      ident = ident && XR[k] == ((k == 0 || k == 4 || k == 8) ? 1.0 : 0.0);
    }
    for (int k = 0; k < 3; ++k) r[TB::XT + 9 + k] = (T)Xt[k];
  }
  T *const sc = t + TB::SC;
  sc[TB::DT] = (T)m->dt;
  // (forward_dynamics.hpp:242: the base acceleration is -gravity in BASE coordinates, "not rotated": in world coordinates
  //  the links feel base_X_world.rot * gravity)
  for (int k = 0; k < 3; ++k) {
    double acc = 0.0;
    for (int c = 0; c < 3; ++c) acc += m->base_X_world_rot[3 * k + c] * m->gravity[c];
    sc[TB::GRAV + k] = (T)acc;
  }
  sc[TB::BASE_R8] = (T)m->base_X_world_rot[8];
  sc[TB::NUM_VISUALS] = (T)(m->pack_visuals ? m->num_visuals : 0);
  sc[TB::PACK_VISUALS] = (T)m->pack_visuals;
  sc[TB::OUTPUT_DIM] = (T)m->output_dim;
  sc[TB::NUM_LINKS] = (T)n;
  sc[TB::XT_IDENT] = ident ? T(1) : T(0);
  {
    bool vid = true;
    for (int v = 0; v < (m->pack_visuals ? m->num_visuals : 0); ++v)
      for (int k = 0; k < 9; ++k) vid = vid && m->visuals[v].X_rot[k] == ((k == 0 || k == 4 || k == 8) ? 1.0 : 0.0);
    sc[TB::VIS_IDENT] = vid ? T(1) : T(0);
  }
  d->chain = n;
}
