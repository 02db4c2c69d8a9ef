// tds_oct_model.h — the constant table of the 8-lane kernel (tds_oct.hip) and the host-side detection of the models it takes.
// Included by tds_device_model.h behind the definition of DevModel; a header of its own so that a change here recompiles
// tds_oct.hip and tds_api.hip only.
#pragma once

// The constants of a two-link-leg star (DevModel::oct) as the 8-lane kernel of tds_oct.hip reads them: ONE flat table of
// compute scalars, built on the host (tds_oct_detect below), copied into LDS at the top of a launch.  Offsets in scalars.
// Per leg-link lane (lane = 2 leg + position, link 6 + lane) a record of LSTR scalars — a stride whose eight records
// start on different LDS banks — then the root body's block, then the model's scalars (integers as exact scalars).
struct TdsOctTab {
  static constexpr int LSTR = 70;
  // the lane's record
  static constexpr int S = 0, XT = 6, MASS = 18, COM = 19, INER = 22, IPOSE = 31, STIFF = 32, DAMP = 33, JT = 34, ACT = 35,
                       CPR0 = 36, CPL0 = 37, CPR1 = 40, CPL1 = 41, VIS = 44,  // VIS: 12 (rotation 9 | translation 3)
                       AXINV = 56,  // 1 / |S_angular|
                       // the joint rotation as Rodrigues' formula R = cos I + sin [n]x + (1 - cos) n n^T about the UNIT axis n (every
                       // revolute type of link.hpp:229-287, the unnormalised REVOLUTE_AXIS included: its axis-angle quaternion is the
                       // rotation about S / |S|): n (3) | n n^T as xx xy xz yy yz zz (6) | 1 for a revolute joint, 0 for a prismatic one
                       NAX = 57, NN = 60, ROTF = 66,
                       // which entries (r, r') of the root's Schur complement this lane sums in its three passes: sum_p (8 r + r') << 6 p
                       SCHUR = 67;
  // the root body's block (link 5)
  static constexpr int ROOT = 8 * LSTR;
  static constexpr int R_MASS = 0, R_COM = 1, R_INER = 4, R_CPR = 13 /* < 0: no sphere */, R_CPL = 14, R_VIS = 17;
  // scalars
  static constexpr int SC = ROOT + 30;
  static constexpr int DT = 0, ACTION_LIMIT = 1, BASE_T = 2, GRAV = 5, PLANE_N = 8, PLANE_C = 11, NB = 12, T1 = 15, T2 = 18, CFM = 21,
                       ERP_OVER_DT = 22, RESTITUTION = 23, FRICTION = 24, BASE_R8 = 25, NUM_VISUALS = 26, REWARD_MODE = 27,
                       PGS_ITERATIONS = 28, PACK_VISUALS = 29, OUTPUT_DIM = 30, INV_DT = 31,
                       XT_IDENT = 32;  // 1: every leg link's X_T rotation is the identity (no R_T R_J product)
  static constexpr int TOTAL = SC + 34;
};

static_assert(TdsOctTab::TOTAL <= TDS_OCT_TAB_CAP, "DevModel::oct_tab is too small for the table");

// sets d->oct (and fills d->oct_tab) where the model is the star the 8-lane kernel is built for — see DevModel::oct.
// ncp: contact points of the model; leg_pd_only: the env step with PD control, every action on a leg joint (star_actuation)
template <typename T>
static void tds_oct_detect(const tds_model_t *m, DevModel<T> *d, int ncp, bool leg_pd_only) {
  d->oct = 0;
  if (d->euler_root && sizeof(T) == 8 && m->num_links == 14 && m->dof_qd == 14 && m->dof_q == 14 && m->has_plane && ncp == 17 &&
      m->action_dim == 8 && m->input_dim == 14 + 14 + 8 + 3 && tds_opt_now(TDS_OPT_OCT) != 0 && leg_pd_only) {
    bool ok = true;
    for (int i = 6; ok && i < 14; ++i) {
      const tds_link_t &l = m->links[i];
      ok = l.parent == ((i & 1) ? i - 1 : 5) && l.joint_type >= TDS_JOINT_PRISMATIC_X && l.joint_type <= TDS_JOINT_REVOLUTE_AXIS &&
           l.qd_index == i && l.q_index == i && d->act_index[i] == i - 6;
      ok = ok && d->cp_link[1 + 2 * (i - 6)] == i && d->cp_link[2 + 2 * (i - 6)] == i;
    }
    ok = ok && d->cp_link[0] == 5;
    ok = ok && (d->num_visuals == 0 || d->num_visuals == 9);
    for (int v = 0; ok && v < d->num_visuals; ++v) ok = d->vis_link[v] == 5 + v;
    ok = ok && (m->reward_mode == TDS_REWARD_NONE || m->reward_mode == TDS_REWARD_ANT || m->reward_mode == TDS_REWARD_LAIKAGO);
    if (ok) {
      using TB = TdsOctTab;
      T *const t = d->oct_tab;
      for (int i = 0; i < TB::TOTAL; ++i) t[i] = T(0);
      for (int ln = 0; ln < 8; ++ln) {
        T *const r = t + ln * TB::LSTR;
        const int li = 6 + ln;
        for (int k = 0; k < 6; ++k) r[TB::S + k] = d->S[k][li];
        for (int k = 0; k < 12; ++k) r[TB::XT + k] = d->X_T[k][li];
        r[TB::MASS] = d->mass[li];
        for (int k = 0; k < 3; ++k) r[TB::COM + k] = d->com[k][li];
        for (int k = 0; k < 9; ++k) r[TB::INER + k] = d->inertia[k][li];
        r[TB::IPOSE] = d->init_pose[li];
        r[TB::STIFF] = d->stiffness[li];
        r[TB::DAMP] = d->damping[li];
        r[TB::JT] = (T)d->joint_type[li];
        {
          const double ax[3] = {(double)d->S[0][li], (double)d->S[1][li], (double)d->S[2][li]};
          const double ax2 = ax[0] * ax[0] + ax[1] * ax[1] + ax[2] * ax[2];
          const int jt = d->joint_type[li];
          const bool revolute = jt >= TDS_JOINT_REVOLUTE_X && jt <= TDS_JOINT_REVOLUTE_AXIS && ax2 > 0.0;
          const double inv = ax2 > 0.0 ? 1.0 / sqrt(ax2) : 0.0;
          r[TB::AXINV] = (T)inv;
          const double n[3] = {revolute ? ax[0] * inv : 0.0, revolute ? ax[1] * inv : 0.0, revolute ? ax[2] * inv : 0.0};
          for (int k = 0; k < 3; ++k) r[TB::NAX + k] = (T)n[k];
          r[TB::NN + 0] = (T)(n[0] * n[0]);
          r[TB::NN + 1] = (T)(n[0] * n[1]);
          r[TB::NN + 2] = (T)(n[0] * n[2]);
          r[TB::NN + 3] = (T)(n[1] * n[1]);
          r[TB::NN + 4] = (T)(n[1] * n[2]);
          r[TB::NN + 5] = (T)(n[2] * n[2]);
          r[TB::ROTF] = revolute ? T(1) : T(0);
          int code = 0;
          for (int pass = 0; pass < 3; ++pass) {
            int e = ln + 8 * pass;
            e = e < 21 ? e : 20;
            int rr = 0;
            for (int k = 1; k < 6; ++k) rr += e >= (k * (k + 1)) / 2 ? 1 : 0;
            const int rp = e - (rr * (rr + 1)) / 2;
            code |= (8 * rr + rp) << (6 * pass);
          }
          r[TB::SCHUR] = (T)code;
        }
        r[TB::ACT] = (T)d->act_index[li];
        for (int e = 0; e < 2; ++e) {
          const int c = 1 + 2 * ln + e;
          r[(e ? TB::CPR1 : TB::CPR0)] = d->cp_radius[c];
          for (int k = 0; k < 3; ++k) r[(e ? TB::CPL1 : TB::CPL0) + k] = d->cp_local[k][c];
        }
        if (d->num_visuals)
          for (int k = 0; k < 12; ++k) r[TB::VIS + k] = d->vis_X[k][1 + ln];
      }
      T *const rt = t + TB::ROOT;
      rt[TB::R_MASS] = d->mass[5];
      for (int k = 0; k < 3; ++k) rt[TB::R_COM + k] = d->com[k][5];
      for (int k = 0; k < 9; ++k) rt[TB::R_INER + k] = d->inertia[k][5];
      rt[TB::R_CPR] = d->cp_radius[0];
      for (int k = 0; k < 3; ++k) rt[TB::R_CPL + k] = d->cp_local[k][0];
      if (d->num_visuals)
        for (int k = 0; k < 12; ++k) rt[TB::R_VIS + k] = d->vis_X[k][0];
      T *const sc = t + TB::SC;
      sc[TB::DT] = d->dt;
      sc[TB::INV_DT] = (T)(1.0 / (double)d->dt);
      {
        bool ident = true;
        for (int li = 6; li < 14; ++li)
          for (int k = 0; k < 9; ++k) ident = ident && d->X_T[k][li] == ((k == 0 || k == 4 || k == 8) ? T(1) : T(0));
        sc[TB::XT_IDENT] = ident ? T(1) : T(0);
      }
      sc[TB::ACTION_LIMIT] = d->action_limit;
      for (int k = 0; k < 3; ++k) {
        sc[TB::BASE_T + k] = d->base_t[k];
        sc[TB::GRAV + k] = d->grav[k];
        sc[TB::PLANE_N + k] = d->plane_n[k];
        sc[TB::NB + k] = d->nb[k];
        sc[TB::T1 + k] = d->t1[k];
        sc[TB::T2 + k] = d->t2[k];
      }
      sc[TB::PLANE_C] = d->plane_c;
      sc[TB::CFM] = d->cfm;
      sc[TB::ERP_OVER_DT] = d->erp_over_dt;
      sc[TB::RESTITUTION] = d->restitution;
      sc[TB::FRICTION] = d->friction;
      sc[TB::BASE_R8] = d->base_R[8];
      sc[TB::NUM_VISUALS] = (T)d->num_visuals;
      sc[TB::REWARD_MODE] = (T)d->reward_mode;
      sc[TB::PGS_ITERATIONS] = (T)d->pgs_iterations;
      sc[TB::PACK_VISUALS] = (T)d->pack_visuals;
      sc[TB::OUTPUT_DIM] = (T)d->output_dim;
    }
    d->oct = ok ? 1 : 0;
  }
}
