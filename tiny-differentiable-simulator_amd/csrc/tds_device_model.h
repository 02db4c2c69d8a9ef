// tds_device_model.h — kernel-side constant model ("DevModel") and its host builder.
//
// The C-ABI blob tds_model_t (include/tds_hip.h) is the flattened tds::MultiBody + World.
// The kernels want the same information re-organised for lane-indexed access:
//   * per-link arrays stored component-major  a[c][link]  so that lane == link reads are coalesced,
//   * tree schedule (level of each link, ancestor masks, (link, ancestor) pair list for CRBA),
//   * contact POINTS (sphere centres in link coordinates) instead of geometries:
//     sphere 1, capsule 2, box 8 points  (reference: src/contact_point.hpp:96-198),
//   * the contact frame of the plane (normal_on_b, two tangents) evaluated once on the host
//     with the reference's own plane_space() quirks (src/mb_constraint_solver.hpp:506-520).
#pragma once
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "tds_hip.h"
#include "tds_options.h"

#define TDS_NL 32                      // lanes of links the kernels take (<= TDS_MAX_LINKS of the blob)
#define TDS_ND 32                      // max dof
#define TDS_NCP TDS_MAX_CONTACTS       // 64 contact points
#define TDS_NV TDS_MAX_VISUALS         // 64
#define TDS_NPAIR (TDS_NL * 12)        // (link, strict ancestor) pairs
#define TDS_NPC TDS_MAX_PAIR_CONTACTS  // contact points between the bodies of a multi-body world (all pairs)
#define TDS_NB TDS_MAX_BODIES          // articulated bodies of a world
#define TDS_NBP ((TDS_NB * (TDS_NB - 1)) / 2)  // body pairs i < j
// internal joint types of the expanded model (never in a tds_model_t handed in by a caller)
#define TDS_JOINT_SPH0 9   // first lane of a spherical joint: X_J = quat_to_matrix(q[0..3]), axis x
#define TDS_JOINT_SPH1 10  // second: identity transform, axis y
#define TDS_JOINT_SPH2 11  // third (the link itself): identity transform, axis z

// capacity of DevModel::oct_tab (the table itself: TdsOctTab, tds_oct_model.h — a header of its own, so that a change of the
// table's layout recompiles tds_oct.hip and tds_api.hip only, not the fifteen translation units of the general kernel)
#define TDS_OCT_TAB_CAP 640

template <typename T>
struct DevModel {
  int num_links, dof_q, dof_qd, num_levels;
  int num_cp, num_visuals, action_dim, input_dim, output_dim;
  int step_mode, has_plane, pgs_iterations, pack_visuals;
  int num_pairs, reward_mode, settle_steps;
  int reset_obs_raw_xy;  // observation of a FORCED reset keeps the base x, y (tds_model_t::reset_obs_raw_xy)
  // floating base (multi_body.hpp:66-78, kinematics.hpp:35-62): links 0..5 of THIS table are six pseudo links
  // (3 angular + 3 linear base dofs, the base body's inertia on link 5) in front of the model's own links;
  // the dofs are numbered joints first (0..nj-1), base last (nj..nj+5), so that the right-looking LDL^T
  // reaches the base block as the Schur complement of the joints (= the articulated inertia of the base).
  // The q / qd RECORD keeps the reference's order: q = [quat xyzw | pos | joints], qd = [omega | v | joints].
  int is_floating, nj;  // nj = number of joint dofs (= dof_qd when the base is fixed)
  // spherical joints (link.hpp:168-176,262-266): the link becomes three lanes — two massless pseudo links and
  // the link itself — that share one frame (quaternion of the joint) and carry the three angular dofs along
  // that frame's x, y, z; internal joint types TDS_JOINT_SPH0/1/2 below.  num_spherical > 0 selects the
  // kernels built for them.  Record indices (the q record has 4 coordinates per spherical joint):
  int num_spherical;
  int q_rec[TDS_NL];    // index of the link's (first) coordinate in the q record, -1: none
  int qd_rec[TDS_NL];   // index of the link's velocity in the qd record, -1: none
  int dof_rec[TDS_ND];  // qd record index of (internal) dof d
  T sph_damping;        // pow(MultiBody::joint_damping_ = 0.995, 1000 dt)   (integrator.hpp:107-112)
  T dt, cfm, erp_over_dt, friction, restitution, action_limit;
  T grav[3];       // base acceleration = -grav (forward_dynamics.hpp:242), world frame
  T base_R[9], base_t[3];
  T plane_n[3], plane_c;
  T nb[3], t1[3], t2[3];  // world_normal_on_b = -n and plane_space(nb)
  // per link ---------------------------------------------------------------------------
  int parent[TDS_NL], level[TDS_NL], joint_type[TDS_NL], q_index[TDS_NL], qd_index[TDS_NL];
  int act_index[TDS_NL];        // PD pose_index of this link or -1 (locomotion_contact_simulation.h:179-257);
                                // -2: lane of a spherical joint whose PD torque is kept (:188-226)
  int sph_q[TDS_NL];            // lanes of a spherical joint: offset of the joint's quaternion in the q record, else -1
  uint32_t anc_dofs[TDS_NL];    // bit d set: dof d lies on the path base -> link (incl. own)
  // chain hand-over of the tree sweeps (lane == link): bit 0: parent == link - 1 inside one 16-lane
  // DPP row -> sweep state travels by DPP row shift; bit 1: link + 1 is such a child of mine;
  // bit 2: I have children that are NOT link + 1 -> they use my per-link LDS record
  int chain_flags[TDS_NL];
  // links with children that are not lane + 1 publish (v, a0) in a side record: its slot, or -1
  int lc_slot[TDS_NL], num_lc_slots;
  // root joint: links 0..root_last form a serial chain from the base whose links 0..root_last-1 are
  // massless with a single child (the 6 "virtual" prismatic/revolute links URDF-derived fixed-base
  // robots carry their free motion on).  The dynamics sweeps then treat joints 0..root_last as ONE
  // (root_last+1)-dof joint of link root_last (one small SPD solve) instead of root_last+1 tree
  // levels.  -1: no such chain.
  int root_last;
  // kinematics only: links 0..kin_chain_last are a serial chain in consecutive lanes (parent of lane i is lane i - 1),
  // massless or not — their world transforms, velocities and bias accelerations are prefix scans over the lanes (4 DPP
  // rounds for up to 16 links) instead of one tree level per link; the level loop then starts at kin_lev0, the first
  // level that holds a link outside the chain.  (>= root_last; a pendulum is all chain.)
  int kin_chain_last, kin_lev0;
  // The root chain of the URDF-derived fixed-base robots in closed form: links 0..5 = prismatic X, Y, Z, revolute X, Y, Z
  // with identity X_T (the Ant, Laikago: Appendix A of SURVEY.md), links 0..4 massless without shapes or visuals.
  // World transform, motion axes, velocity and bias acceleration of link 5 then follow from six broadcasts of
  // (q, sin q, cos q, qd) without any scan over the chain.  0: no; 1: yes (the base frame's rotation is the identity too).
  int euler_root;
  // ... and behind it nothing but serial chains of equal length leg_len = 2 or 4 in consecutive lanes, every one hanging
  // off link 5 (the Ant's hip + ankle, Laikago's hip + upper + lower + toe): their kinematics is one segmented prefix scan
  // over the lanes (log2(leg_len) rounds of DPP shifts) instead of leg_len tree levels.  0: no.
  int leg_len;
  // 1: link i carries dof i for every link (lane == link == dof: the Ant; not Laikago, whose fixed toes own no dof) —
  // per-link results that feed per-dof computations then stay in the lane's registers instead of crossing through LDS
  int dof_identity;
  // 1: the model is the STAR the 16-lane kernel of tds_quad.hip is built for: the closed-form root chain (euler_root), behind
  // it exactly four legs of four links each in consecutive lanes — three 1-dof joints and a FIXED toe —, plane contacts
  // on the toes only (one sphere each, in leg order), visuals on links 5 .. 21 in link order, unactuated root links without
  // joint springs.  Laikago (BASELINE config 4) is one; option quad = 0 keeps such a model on the general kernel.
  int quad;
  // 1: the model is the STAR the 8-lane kernel of tds_oct.hip is built for: the closed-form root chain, behind it exactly four
  // legs of two links each in consecutive lanes, every leg link with a 1-dof joint, a PD actuator and one capsule, the root
  // body with one sphere (17 contact points in the reference's order), visuals on links 5 .. 13 in link order, env step
  // with PD control.  The gym Ant (BASELINE configs 3 and 5) is one; option oct = 0 keeps such a model on the general kernel.
  int oct;
  // n > 0: the model is a fixed-base serial CHAIN of n links (2 .. 8) without contacts, torques given directly — the kernel of
  // tds_chain.hip (tds_chain_model.h: tds_chain_detect); BASELINE configs 1 and 2 (cartpole, pendulum5).  Option chain = 0
  // keeps such a model on the general kernel.
  int chain;
  T oct_tab[TDS_OCT_TAB_CAP];  // (TdsOctTab, tds_oct_model.h, where oct == 1; TdsChainTab, tds_chain_model.h, where chain > 0)
  T X_T[12][TDS_NL];            // rot (row-major 9) | trans (3)
  T S[6][TDS_NL];
  T mass[TDS_NL], com[3][TDS_NL], inertia[9][TDS_NL];
  T stiffness[TDS_NL], damping[TDS_NL], init_pose[TDS_NL];
  // per dof ----------------------------------------------------------------------------
  int dof_link[TDS_ND];
  T reset_q[TDS_ND], reset_noise[TDS_ND];
  // contact points ---------------------------------------------------------------------
  int cp_link[TDS_NCP];
  T cp_local[3][TDS_NCP], cp_radius[TDS_NCP];
  // visuals ----------------------------------------------------------------------------
  int vis_link[TDS_NV];
  T vis_X[12][TDS_NV];
  // CRBA off-diagonal work list: M[qd(i)][qd(j)] for j a strict ancestor of i, both with a dof
  int16_t pair_i[TDS_NPAIR], pair_j[TDS_NPAIR];
  // multi-body worlds (tds_model_t::num_bodies >= 2; kernels of KIND 3) ------------------------------
  int num_bodies;               // >= 2: links with body_of_link == b hang off the base of body b; 0: one body
  int body_dof0[TDS_NB + 1];    // dofs body_dof0[b] .. body_dof0[b + 1] - 1 belong to body b
  int body_of_link[TDS_NL];
  T base_Rb[TDS_NB][9], base_tb[TDS_NB][3], gravb[TDS_NB][3];  // per body (entry 0 repeats base_R, base_t, grav)
  // body pairs a < b in the reference's order (world.hpp:212-216: 0-1, 0-2, .., 1-2, ..); each is a contact pass of its
  // own; the contact points of pair p are pc[pair_pc0[p] .. pair_pc0[p + 1] - 1]
  int num_bpairs, bpair_a[TDS_NBP], bpair_b[TDS_NBP], bpair_pc0[TDS_NBP + 1];
  // ... with FLOATING bases among the bodies (kernels of KIND 4): a floating body's six pseudo links sit in front of its
  // links, its dofs are numbered joints first, base last (see is_floating above), body by body
  int multi_floating;           // 1: some body of the world has a floating base
  int fb_k[TDS_NL];             // lane is pseudo link k = 0..5 of a floating body's base; -1: an ordinary link
  int fb_q[TDS_NL];             // pseudo links: index of the body's quaternion in the q record (position at + 4)
  int tau_rec[TDS_NL];          // TAU mode: index of the link's torque within the action part of the record, -1: none
  int dof_body[TDS_ND];         // body of dof d
  int dof_joint0[TDS_ND];       // first (joint) dof of that body
  int dof_base0[TDS_ND];        // first of the six base dofs of that body; -1: its base is fixed
  int dof_fbq[TDS_ND];          // d is a BASE dof: index of the body's quaternion in the q record; else -1
  // contact points between a geometry of body a and a geometry of body b, in the reference's order (world.hpp:206-282:
  // geoms of a outer, geoms of b inner; capsule-sphere: +L/2 end, then -L/2 end).  Each is a pair of spheres:
  int num_pc;
  int pc_link_a[TDS_NPC], pc_link_b[TDS_NPC];  // owning links (global index; -1: the body's base)
  int pc_swap[TDS_NPC];         // 1: the dispatcher ran the pair with swapped arguments (sphere of A, capsule of B:
                                //    contact_point.hpp:478-495) — same contact, different rounding of the two points
  T pc_loc_a[3][TDS_NPC], pc_loc_b[3][TDS_NPC], pc_rad_a[TDS_NPC], pc_rad_b[TDS_NPC];
};

#include "tds_oct_model.h"
#include "tds_chain_model.h"

// reference: src/mb_constraint_solver.hpp:506-520 (incl. k = sqrt(a) and p[2] quirks)
static inline void tds_plane_space(const double *n, double *p, double *q) {
  double n_sqr = n[2] * n[2];
  int gt = n_sqr > 0.5;
  double a = n[1] * n[1] + (gt ? n_sqr : n[0] * n[0]);
  double k = sqrt(a);
  p[0] = gt ? 0.0 : -n[1] * k;
  p[1] = gt ? -n[2] * k : n[0] * k;
  p[2] = n[1] * k;
  q[0] = gt ? a * k : -n[2] * p[1];
  q[1] = gt ? -n[0] * p[2] : n[2] * p[0];
  q[2] = gt ? n[0] * p[1] : a * k;
}

// the expanded form of a model with a floating base and/or spherical joints (tds_expand_model below)
struct TdsExpanded {
  tds_model_t m;        // links incl. pseudo links, internal dof numbering in q_index == qd_index
  int q_rec[TDS_NL], qd_rec[TDS_NL];
  int pd_on[TDS_NL];    // the PD loop of the env step visits this link (locomotion_contact_simulation.h:180-181);
                        // spherical lanes: 2 = its torque is stored (floating base or link index >= 4, :215-221), 1 = dropped
  int sph_q[TDS_NL];    // spherical lanes: offset of the joint's quaternion in the q record (all three lanes)
  int num_spherical;
  // worlds of several bodies with floating bases among them (tds_expand_multibody):
  int multi;            // 1: expanded by tds_expand_multibody
  int body_fl[TDS_NB];  // body b has a floating base
  int fb_k[TDS_NL], fb_q[TDS_NL], tau_rec[TDS_NL];   // see DevModel
};

// `ex`: m == &ex->m is an EXPANDED model; NULL: a plain fixed-base model of 1-dof joints.
template <typename T>
static int tds_build_dev_model_impl(const tds_model_t *m, DevModel<T> *d, char *why, const TdsExpanded *ex) {
  memset(d, 0, sizeof(*d));
  why[0] = 0;
  const bool exm = ex != nullptr && ex->multi != 0;  // several bodies, floating bases among them
  const bool fl = ex != nullptr && !exm && m->is_floating != 0;
  const int nsph = ex ? ex->num_spherical : 0;
  int nflb = 0;  // floating bodies of a multi-body world
  for (int b = 0; exm && b < m->num_bodies && b < TDS_NB; ++b) nflb += ex->body_fl[b] ? 1 : 0;
#define TDS_FAIL(code, msg)          \
  do {                               \
    strncpy(why, msg, 127);          \
    why[127] = 0;                    \
    return code;                     \
  } while (0)
  auto fixed_joint = [](const tds_link_t &l) { return l.joint_type == TDS_JOINT_FIXED; };
  if (m->abi_version != TDS_HIP_ABI_VERSION) TDS_FAIL(TDS_ERR_INVALID_ARG, "model abi_version mismatch");
  if (m->num_links < 1 || m->num_links > TDS_NL) TDS_FAIL(TDS_ERR_INVALID_ARG, "num_links out of range");
  if (m->dof_qd < 1 || m->dof_qd > TDS_ND || m->dof_q != m->dof_qd + (fl ? 1 : 0) + nflb + nsph || m->dof_q > TDS_ND)
    TDS_FAIL(TDS_ERR_UNSUPPORTED, "dof out of range (<= 32 velocities, <= 32 coordinates) or dof_q inconsistent with the joints");
  if (nsph && m->reward_mode != TDS_REWARD_NONE && m->reward_mode != TDS_REWARD_HUMANOID)
    TDS_FAIL(TDS_ERR_UNSUPPORTED, "the Ant / Laikago reward rules read a 1-dof-joint state record");
  if (m->reward_mode == TDS_REWARD_HUMANOID &&
      !(nsph && !fl && m->num_links > 3 && m->links[3].joint_type == TDS_JOINT_SPH0 && ex->q_rec[3] == 3))
    TDS_FAIL(TDS_ERR_UNSUPPORTED, "the humanoid reward rule needs the xyz + spherical root joint (quaternion at q[3..6])");
  d->num_spherical = nsph;
  d->sph_damping = (T)pow(0.995, 1000.0 * m->dt);
  if (fl && m->reward_mode != TDS_REWARD_NONE)
    TDS_FAIL(TDS_ERR_UNSUPPORTED, "the Ant / Laikago reward rules read a fixed-base state record");
  d->is_floating = fl ? 1 : 0;
  d->nj = fl ? m->dof_qd - 6 : m->dof_qd - 6 * nflb;
  d->multi_floating = exm ? 1 : 0;
  if (m->num_geoms < 0 || m->num_geoms > TDS_MAX_GEOMS) TDS_FAIL(TDS_ERR_INVALID_ARG, "num_geoms out of range");
  if (m->num_visuals < 0 || m->num_visuals > TDS_NV) TDS_FAIL(TDS_ERR_INVALID_ARG, "num_visuals out of range");
  if (m->step_mode != TDS_STEP_LOCOMOTION && m->step_mode != TDS_STEP_TAU)
    TDS_FAIL(TDS_ERR_INVALID_ARG, "unknown step_mode");
  if (m->pgs_iterations < 1) TDS_FAIL(TDS_ERR_INVALID_ARG, "pgs_iterations < 1");
  if (m->action_dim < 0 || m->action_dim > TDS_MAX_ACTIONS) TDS_FAIL(TDS_ERR_INVALID_ARG, "action_dim out of range (0..TDS_MAX_ACTIONS)");
  d->num_links = m->num_links;
  d->dof_q = m->dof_q;
  d->dof_qd = m->dof_qd;
  d->num_visuals = m->pack_visuals ? m->num_visuals : 0;
  d->action_dim = m->action_dim;
  d->input_dim = m->input_dim;
  d->output_dim = m->output_dim;
  d->step_mode = m->step_mode;
  d->has_plane = m->has_plane;
  d->pgs_iterations = m->pgs_iterations;
  d->pack_visuals = m->pack_visuals;
  d->reward_mode = m->reward_mode;
  d->settle_steps = m->settle_steps < 0 ? 0 : m->settle_steps;
  d->reset_obs_raw_xy = m->reset_obs_raw_xy != 0;
  for (int k = 0; k < TDS_ND && k < TDS_MAX_DOF; ++k) {
    d->reset_q[k] = (T)m->reset_q[k];
    d->reset_noise[k] = (T)m->reset_noise[k];
  }
  const int nq = m->dof_q, nd = m->dof_qd;
  const int need_in = nq + nd + m->action_dim + (m->step_mode == TDS_STEP_LOCOMOTION ? 3 : 0);
  if (m->input_dim < need_in) TDS_FAIL(TDS_ERR_INVALID_ARG, "input_dim too small for [q|qd|action|vars]");
  const int need_out = nq + nd + (m->pack_visuals ? 7 * m->num_visuals + 1 : 0);
  if (m->output_dim < need_out) TDS_FAIL(TDS_ERR_INVALID_ARG, "output_dim too small for [q|qd|visuals|up]");
  if (m->step_mode == TDS_STEP_TAU && m->action_dim != d->nj)
    TDS_FAIL(TDS_ERR_INVALID_ARG, "TAU mode needs action_dim == number of joint dofs (dof_actuated)");
  d->dt = (T)m->dt;
  d->cfm = (T)m->cfm;
  d->erp_over_dt = (T)(m->erp / m->dt);
  d->friction = (T)m->friction;
  d->restitution = (T)m->restitution;
  d->action_limit = (T)m->action_limit;
  // base acceleration is -gravity taken in BASE coordinates (rbdl_convention = false,
  // forward_dynamics.hpp:237-243); the kernels work in world coordinates, so rotate it.
  for (int r = 0; r < 3; ++r) {
    double g = 0;
    for (int c = 0; c < 3; ++c) g += m->base_X_world_rot[3 * r + c] * m->gravity[c];
    // floating base: the WORLD components of gravity are added to the base acceleration as they are
    // (forward_dynamics.hpp:315-319)
    d->grav[r] = (T)(fl ? m->gravity[r] : g);
  }
  for (int k = 0; k < 9; ++k) d->base_R[k] = (T)m->base_X_world_rot[k];
  for (int k = 0; k < 3; ++k) d->base_t[k] = (T)m->base_X_world_trans[k];
  double nb[3], t1[3], t2[3];
  for (int k = 0; k < 3; ++k) {
    d->plane_n[k] = (T)m->plane_normal[k];
    nb[k] = -m->plane_normal[k];
  }
  d->plane_c = (T)m->plane_constant;
  tds_plane_space(nb, t1, t2);
  for (int k = 0; k < 3; ++k) {
    d->nb[k] = (T)nb[k];
    d->t1[k] = (T)t1[k];
    d->t2[k] = (T)t2[k];
  }
  const int NBod = m->num_bodies;
  const bool two = NBod >= 2;  // a world of several articulated bodies
  if (NBod > TDS_NB || NBod < 0) TDS_FAIL(TDS_ERR_UNSUPPORTED, "worlds of more than TDS_MAX_BODIES articulated bodies");
  int body_l0[TDS_NB + 1] = {0}, body_g0[TDS_NB + 1] = {0};  // first link / geom of each body (+ end)
  if (two) {
    if ((ex != nullptr && !exm) || (!exm && m->is_floating))
      TDS_FAIL(TDS_ERR_UNSUPPORTED, "multi-body worlds: 1-dof joints (floating bases through tds_expand_multibody)");
    if (m->step_mode != TDS_STEP_TAU) TDS_FAIL(TDS_ERR_UNSUPPORTED, "multi-body worlds step in TAU mode");
    if (m->reward_mode != TDS_REWARD_NONE) TDS_FAIL(TDS_ERR_UNSUPPORTED, "multi-body worlds carry no reward rule");
    d->num_bodies = NBod;
    for (int b = 1; b < NBod; ++b) {
      const tds_body_t &B = m->bodies[b];
      if (B.is_floating && !exm) TDS_FAIL(TDS_ERR_UNSUPPORTED, "multi-body worlds: a floating base needs the expanded model");
      if (B.first_link <= body_l0[b - 1] || B.first_link >= m->num_links || B.first_geom < body_g0[b - 1] ||
          B.first_geom > m->num_geoms)
        TDS_FAIL(TDS_ERR_INVALID_ARG, "bodies[].first_link / first_geom out of range or not ascending");
      body_l0[b] = B.first_link;
      body_g0[b] = B.first_geom;
    }
    body_l0[NBod] = m->num_links;
    body_g0[NBod] = m->num_geoms;
    for (int b = 0; b < NBod; ++b) {
      const double *R = b == 0 ? m->base_X_world_rot : m->bodies[b].base_X_world_rot;
      const double *t = b == 0 ? m->base_X_world_trans : m->bodies[b].base_X_world_trans;
      for (int r = 0; r < 3; ++r) {
        double g = 0;
        for (int c = 0; c < 3; ++c) g += R[3 * r + c] * m->gravity[c];
        // (a floating body: the WORLD components of gravity, added to its base acceleration as they are,
        //  forward_dynamics.hpp:315-319)
        d->gravb[b][r] = (T)((exm && ex->body_fl[b]) ? m->gravity[r] : g);
        d->base_tb[b][r] = (T)t[r];
      }
      for (int k = 0; k < 9; ++k) d->base_Rb[b][k] = (T)R[k];
    }
  }
  // links
  int max_level = 0, pose_index = 0, ndof = 0;
  uint32_t anc_links[TDS_NL];
  for (int i = 0; i < m->num_links; ++i) {
    const tds_link_t &l = m->links[i];
    if (l.parent >= i || l.parent < -1) TDS_FAIL(TDS_ERR_INVALID_ARG, "links must be ordered parent-before-child");
    const bool sph_lane = ex != nullptr && l.joint_type >= TDS_JOINT_SPH0 && l.joint_type <= TDS_JOINT_SPH2;
    if (!sph_lane && (l.joint_type == TDS_JOINT_SPHERICAL || l.joint_type < TDS_JOINT_FIXED || l.joint_type > TDS_JOINT_SPHERICAL))
      TDS_FAIL(TDS_ERR_UNSUPPORTED, "unknown joint type");
    d->q_rec[i] = ex ? ex->q_rec[i] : (l.joint_type == TDS_JOINT_FIXED ? -1 : l.q_index);
    d->qd_rec[i] = ex ? ex->qd_rec[i] : (l.joint_type == TDS_JOINT_FIXED ? -1 : l.qd_index);
    d->parent[i] = l.parent;
    {
      int bo = 0;
      while (two && bo + 1 < NBod && i >= body_l0[bo + 1]) ++bo;
      d->body_of_link[i] = bo;
      if (two && i == body_l0[bo]) d->body_dof0[bo] = ndof;
    }
    if (two && l.parent >= 0 && d->body_of_link[l.parent] != d->body_of_link[i])
      TDS_FAIL(TDS_ERR_INVALID_ARG, "a link's parent belongs to another body");
    if (two && (l.joint_type == TDS_JOINT_SPHERICAL)) TDS_FAIL(TDS_ERR_UNSUPPORTED, "multi-body worlds: 1-dof joints");
    d->level[i] = l.parent >= 0 ? d->level[l.parent] + 1 : 0;
    if (d->level[i] > max_level) max_level = d->level[i];
    d->joint_type[i] = l.joint_type;
    const bool fixed = l.joint_type == TDS_JOINT_FIXED;
    if (!fixed) {
      if (ex) {  // internal numbering of the expanded model: checked by tds_expand_model
        if (l.qd_index < 0 || l.qd_index >= nd) TDS_FAIL(TDS_ERR_INVALID_ARG, "qd index out of range");
        d->dof_link[l.qd_index] = i;
        ++ndof;
      } else {
        if (l.q_index != ndof || l.qd_index != ndof) TDS_FAIL(TDS_ERR_INVALID_ARG, "q/qd indices must be dense in link order");
        d->dof_link[ndof++] = i;
      }
    }
    d->q_index[i] = fixed ? -1 : l.q_index;
    d->qd_index[i] = fixed ? -1 : l.qd_index;
    d->anc_dofs[i] = (l.parent >= 0 ? d->anc_dofs[l.parent] : 0u) | (fixed ? 0u : (1u << l.qd_index));
    anc_links[i] = (l.parent >= 0 ? anc_links[l.parent] | (1u << l.parent) : 0u);
    d->act_index[i] = -1;
    d->sph_q[i] = sph_lane ? ex->sph_q[i] : -1;
    d->fb_k[i] = exm ? ex->fb_k[i] : -1;
    d->fb_q[i] = exm ? ex->fb_q[i] : -1;
    d->tau_rec[i] = exm ? ex->tau_rec[i] : (fixed_joint(l) ? -1 : l.qd_index);
    const bool pd_here = ex ? ex->pd_on[i] != 0 : i >= m->pd_start_link;
    if (m->step_mode == TDS_STEP_LOCOMOTION && pd_here && sph_lane) {
      // the PD block's spherical branch (locomotion_contact_simulation.h:188-226): four pose slots per joint (:223),
      // no action; lanes whose torque is kept carry act_index = -2
      if (l.joint_type == TDS_JOINT_SPH0) pose_index += 4;
      if (ex->pd_on[i] == 2) d->act_index[i] = -2;
    } else if (m->step_mode == TDS_STEP_LOCOMOTION && pd_here && !fixed) {
      if (pose_index >= m->action_dim) TDS_FAIL(TDS_ERR_INVALID_ARG, "more PD links than action_dim");
      d->act_index[i] = pose_index;
      d->init_pose[i] = (T)m->initial_poses[pose_index];
      ++pose_index;
    }
    for (int k = 0; k < 9; ++k) d->X_T[k][i] = (T)l.X_T_rot[k];
    for (int k = 0; k < 3; ++k) d->X_T[9 + k][i] = (T)l.X_T_trans[k];
    for (int k = 0; k < 6; ++k) d->S[k][i] = (T)l.S[k];
    d->mass[i] = (T)l.mass;
    for (int k = 0; k < 3; ++k) d->com[k][i] = (T)l.com[k];
    for (int k = 0; k < 9; ++k) d->inertia[k][i] = (T)l.inertia[k];
    d->stiffness[i] = (T)l.stiffness;
    d->damping[i] = (T)l.damping;
  }
  if (ndof != nd) TDS_FAIL(TDS_ERR_INVALID_ARG, "dof_qd does not match the joints");
  if (two) d->body_dof0[NBod] = ndof;
  for (int dd = 0; dd < TDS_ND; ++dd) d->dof_base0[dd] = d->dof_fbq[dd] = -1;
  if (two) {
    for (int b = 0; b < NBod; ++b) {
      const bool bf = exm && ex->body_fl[b];
      for (int dd = d->body_dof0[b]; dd < d->body_dof0[b + 1]; ++dd) {
        d->dof_body[dd] = b;
        d->dof_joint0[dd] = d->body_dof0[b];
        d->dof_base0[dd] = bf ? d->body_dof0[b + 1] - 6 : -1;
        const int lk = d->dof_link[dd];
        d->dof_fbq[dd] = (bf && d->fb_k[lk] >= 0) ? d->fb_q[lk] : -1;
      }
    }
  }
  for (int dd = 0; dd < nd; ++dd) d->dof_rec[dd] = d->qd_rec[d->dof_link[dd]];
  {
    // TDS_HIP_NO_CHAIN=1 sends every parent/child hand-over through LDS (A/B testing of the two paths)
    const bool use_chain = !tds_opt_now_flag(TDS_OPT_NO_CHAIN);
    for (int i = 0; i < m->num_links; ++i) {
      const int par = m->links[i].parent;
      if (par < 0) continue;
      if (use_chain && par == i - 1 && (i % 16) != 0) {
        d->chain_flags[i] |= 1;
        d->chain_flags[par] |= 2;
      } else {
        d->chain_flags[par] |= 4;
      }
    }
    for (int i = 0; i < m->num_links; ++i) d->lc_slot[i] = (d->chain_flags[i] & 4) ? d->num_lc_slots++ : -1;
    d->root_last = -1;
    {
      const bool no_rootjoint = tds_opt_now_flag(TDS_OPT_NO_ROOTJOINT);
      int nchild[TDS_NL] = {0}, nroots = 0;
      for (int i = 0; i < m->num_links; ++i) {
        if (m->links[i].parent >= 0) nchild[m->links[i].parent]++;
        else ++nroots;
      }
      auto massless = [&](int i) {
        const tds_link_t &l = m->links[i];
        if (l.mass != 0.0) return false;
        for (int c = 0; c < 9; ++c)
          if (l.inertia[c] != 0.0) return false;
        return true;
      };
      const bool ok = use_chain && !no_rootjoint && nroots == 1;
      int k = 0;  // first link of the root chain that is not (massless, single child, movable)
      // (the first two lanes of a spherical joint may END the chain but never lie inside it: their
      //  velocity-product acceleration is not the chain's prefix sum, see phase C of the kernels)
      while (ok && k < m->num_links - 1 && k < 5 && massless(k) && nchild[k] == 1 && m->links[k + 1].parent == k &&
             m->links[k].joint_type != TDS_JOINT_FIXED && m->links[k].joint_type != TDS_JOINT_SPH0 &&
             m->links[k].joint_type != TDS_JOINT_SPH1)
        ++k;
      if (ok && k >= 1 && m->links[k].joint_type != TDS_JOINT_FIXED && m->links[k].joint_type != TDS_JOINT_SPH1 &&
          m->links[k].joint_type != TDS_JOINT_SPH2)
        d->root_last = k;
      if (fl) d->root_last = 5;  // the six pseudo links ARE the root joint (the kernels special-case their kinematics)
      d->kin_chain_last = d->root_last;
      const bool no_kinchain = tds_opt_now_flag(TDS_OPT_NO_KINCHAIN);
      if (use_chain && !fl && d->num_spherical == 0 && d->num_bodies < 2 && nroots == 1 && !no_kinchain &&
          m->num_links > 0 && m->links[0].parent < 0) {
        int c = 0;
        while (c + 1 < m->num_links && c + 1 < 16 && m->links[c + 1].parent == c && (d->chain_flags[c + 1] & 1)) ++c;
        // a longer chain costs scan rounds (ceil(log2(c + 1)), ~0.8 k cycles each) and saves tree levels (~1.7 k each)
        // only if no link outside it sits at one of its levels: the Ant's and Laikago's first leg would extend the
        // chain without saving a level (the other legs still need theirs) — Laikago would even pay a fourth round
        auto cost = [&](int last) {
          int rounds = 0;
          while ((1 << rounds) <= last) ++rounds;
          unsigned levels = 0;  // levels that hold a link outside the chain
          for (int i = last + 1; i < m->num_links; ++i) levels |= 1u << (d->level[i] & 31);
          return 800 * rounds + 1700 * __builtin_popcount(levels);
        };
        if (c > d->kin_chain_last && cost(c) < cost(d->kin_chain_last)) d->kin_chain_last = c;
      }
    }
  }
  d->euler_root = 0;
  {
    bool ok = !tds_opt_now_flag(TDS_OPT_NO_EULERROOT) && !fl && d->num_spherical == 0 && d->num_bodies < 2 && d->root_last == 5 &&
              d->kin_chain_last == 5 && m->num_links > 6;
    static const int want[6] = {TDS_JOINT_PRISMATIC_X, TDS_JOINT_PRISMATIC_Y, TDS_JOINT_PRISMATIC_Z,
                                TDS_JOINT_REVOLUTE_X,  TDS_JOINT_REVOLUTE_Y,  TDS_JOINT_REVOLUTE_Z};
    for (int i = 0; ok && i < 6; ++i) {
      const tds_link_t &l = m->links[i];
      ok = l.joint_type == want[i] && l.parent == i - 1 && l.qd_index == i && l.q_index == i;
      for (int c = 0; ok && c < 9; ++c) ok = l.X_T_rot[c] == ((c == 0 || c == 4 || c == 8) ? 1.0 : 0.0);
      for (int c = 0; ok && c < 3; ++c) ok = l.X_T_trans[c] == 0.0;
      for (int c = 0; ok && c < 6; ++c) ok = l.S[c] == ((c == (i < 3 ? 3 + i : i - 3)) ? 1.0 : 0.0);
      ok = ok && l.stiffness == 0.0 && l.damping == 0.0;
    }
    // (the closed form leaves links 0..4 without a world transform of their own: nothing may hang on them)
    for (int g = 0; ok && g < m->num_geoms; ++g) ok = !(m->geoms[g].link >= 0 && m->geoms[g].link < 5);
    for (int v = 0; ok && v < m->num_visuals; ++v) ok = !(m->visuals[v].link >= 0 && m->visuals[v].link < 5);
    if (ok) {
      bool ident = true;
      for (int c = 0; c < 9; ++c) ident = ident && m->base_X_world_rot[c] == ((c == 0 || c == 4 || c == 8) ? 1.0 : 0.0);
      d->euler_root = ident ? 1 : 0;  // (a rotated base frame: the general scan)
    }
    d->leg_len = 0;
    // (double arithmetic only: in the pure float build the re-associated products of the scan cost Laikago a quarter of
    //  a digit — 4.7e-4 against 3.5e-4 of the level loop, the reference's own float path: 1.1e-4; tests/test_f32.py)
    if (d->euler_root && sizeof(T) == 8 && !tds_opt_now_flag(TDS_OPT_NO_LEGSCAN)) {
      const int nleg = m->num_links - 6;
      for (int len = 4; len >= 2 && d->leg_len == 0; len >>= 1) {
        if (nleg < len || nleg % len != 0) continue;
        bool legs = true;
        for (int i = 6; legs && i < m->num_links; ++i) {
          const int j = (i - 6) % len;
          legs = m->links[i].parent == (j == 0 ? 5 : i - 1);
        }
        if (legs) d->leg_len = len;
      }
    }
  }
  d->dof_identity = (!fl && d->num_spherical == 0 && d->num_bodies < 2 && m->num_links == m->dof_qd) ? 1 : 0;
  for (int i = 0; i < m->num_links && d->dof_identity; ++i)
    if (m->links[i].qd_index != i) d->dof_identity = 0;
  d->num_levels = max_level + 1;
  d->kin_lev0 = d->num_levels;
  for (int i = d->kin_chain_last + 1; i < m->num_links; ++i)
    if (d->level[i] < d->kin_lev0) d->kin_lev0 = d->level[i];
  // CRBA pair list
  int np = 0;
  for (int i = 0; i < m->num_links; ++i) {
    if (d->qd_index[i] < 0) continue;
    for (int j = 0; j < i; ++j)
      if ((anc_links[i] >> j) & 1u) {
        if (d->qd_index[j] < 0) continue;
        if (np >= TDS_NPAIR) TDS_FAIL(TDS_ERR_UNSUPPORTED, "kinematic tree too deep for the CRBA pair list");
        d->pair_i[np] = (int16_t)i;
        d->pair_j[np] = (int16_t)j;
        ++np;
      }
  }
  d->num_pairs = np;
  // contact points (only generated when a plane is present: world.hpp:206-282 pairs bodies)
  int ncp = 0;
  if (m->has_plane) {
    for (int g = 0; g < m->num_geoms; ++g) {
      const tds_geom_t &G = m->geoms[g];
      if (G.link < (two ? -NBod : -1) || G.link >= m->num_links) TDS_FAIL(TDS_ERR_INVALID_ARG, "geom link out of range");
      int npts = 0;
      double off[8][3];
      double radius = G.radius;
      if (G.type == TDS_GEOM_SPHERE) {
        npts = 1;
        off[0][0] = off[0][1] = off[0][2] = 0;
      } else if (G.type == TDS_GEOM_CAPSULE) {  // contact_point.hpp:127-161
        npts = 2;
        for (int e = 0; e < 2; ++e) {
          off[e][0] = off[e][1] = 0;
          off[e][2] = (e == 0 ? 0.5 : -0.5) * G.length;
        }
      } else if (G.type == TDS_GEOM_BOX) {  // contact_point.hpp:163-198, geometry.hpp:244-259
        npts = 8;
        radius = G.radius > 1e-2 ? G.radius : 1e-2;
        double dx = G.extents[0] * 0.5 - radius, dy = G.extents[1] * 0.5 - radius, dz = G.extents[2] * 0.5 - radius;
        for (int c = 0; c < 8; ++c) {
          off[c][0] = (c & 4) ? -dx : dx;
          off[c][1] = (c & 2) ? -dy : dy;
          off[c][2] = (c & 1) ? -dz : dz;
        }
      } else {
        continue;  // meshes etc. are ignored by the reference too (urdf_to_multi_body.hpp:273-274)
      }
      for (int p = 0; p < npts; ++p) {
        if (ncp >= TDS_NCP) TDS_FAIL(TDS_ERR_UNSUPPORTED, "too many contact points");
        d->cp_link[ncp] = G.link;
        d->cp_radius[ncp] = (T)radius;
        for (int r = 0; r < 3; ++r) {
          double v = G.X_trans[r];
          for (int c = 0; c < 3; ++c) v += G.X_rot[3 * r + c] * off[p][c];
          d->cp_local[r][ncp] = (T)v;
        }
        ++ncp;
      }
    }
  }
  d->num_cp = ncp;
  // contact points between the two bodies: every (geometry of A, geometry of B) pair the reference's dispatcher knows
  int npc = 0;
  if (two) {
    auto ends = [&](const tds_geom_t &G, double out[2][3]) {  // sphere centres of the geometry in its link's frame
      const int n = G.type == TDS_GEOM_CAPSULE ? 2 : 1;
      for (int e = 0; e < n; ++e) {
        const double off[3] = {0.0, 0.0, G.type == TDS_GEOM_CAPSULE ? (e == 0 ? 0.5 : -0.5) * G.length : 0.0};
        for (int r = 0; r < 3; ++r) {
          double v = G.X_trans[r];
          for (int c = 0; c < 3; ++c) v += G.X_rot[3 * r + c] * off[c];
          out[e][r] = v;
        }
      }
      return n;
    };
    int np = 0;
    for (int ba = 0; ba < NBod; ++ba)
      for (int bb = ba + 1; bb < NBod; ++bb) {
        d->bpair_a[np] = ba;
        d->bpair_b[np] = bb;
        d->bpair_pc0[np] = npc;
        for (int ga = body_g0[ba]; ga < body_g0[ba + 1]; ++ga)
          for (int gb = body_g0[bb]; gb < body_g0[bb + 1]; ++gb) {
            const tds_geom_t &A = m->geoms[ga], &B = m->geoms[gb];
            const bool ss = A.type == TDS_GEOM_SPHERE && B.type == TDS_GEOM_SPHERE;
            const bool cs = A.type == TDS_GEOM_CAPSULE && B.type == TDS_GEOM_SPHERE;   // contact_capsule_sphere
            const bool sc = A.type == TDS_GEOM_SPHERE && B.type == TDS_GEOM_CAPSULE;   // ... through the dispatcher's swap
            if (!ss && !cs && !sc) continue;  // (capsule-capsule, boxes, meshes: no function in the dispatcher)
            auto owned = [&](const tds_geom_t &G, int b) {
              return G.link < 0 ? G.link == -1 - b : (G.link >= body_l0[b] && G.link < body_l0[b + 1]);
            };
            if (!owned(A, ba) || !owned(B, bb)) TDS_FAIL(TDS_ERR_INVALID_ARG, "geometry listed under the wrong body");
            double ea[2][3], eb[2][3];
            const int na = ends(A, ea), nb = ends(B, eb);
            for (int p = 0; p < (na > nb ? na : nb); ++p) {
              if (npc >= TDS_NPC) TDS_FAIL(TDS_ERR_UNSUPPORTED, "too many contact points between the bodies");
              d->pc_link_a[npc] = A.link < 0 ? -1 : A.link;
              d->pc_link_b[npc] = B.link < 0 ? -1 : B.link;
              d->pc_swap[npc] = sc ? 1 : 0;
              d->pc_rad_a[npc] = (T)A.radius;
              d->pc_rad_b[npc] = (T)B.radius;
              for (int r = 0; r < 3; ++r) {
                d->pc_loc_a[r][npc] = (T)ea[na == 2 ? p : 0][r];
                d->pc_loc_b[r][npc] = (T)eb[nb == 2 ? p : 0][r];
              }
              ++npc;
            }
          }
        ++np;
      }
    d->num_bpairs = np;
    d->bpair_pc0[np] = npc;
  }
  d->num_pc = npc;
  for (int v = 0; v < d->num_visuals; ++v) {
    const tds_visual_t &V = m->visuals[v];
    if (V.link < 0 || V.link >= m->num_links) TDS_FAIL(TDS_ERR_INVALID_ARG, "visual link out of range");
    d->vis_link[v] = V.link;
    for (int k = 0; k < 9; ++k) d->vis_X[k][v] = (T)V.X_rot[k];
    for (int k = 0; k < 3; ++k) d->vis_X[9 + k][v] = (T)V.X_trans[k];
  }
  // the STARS of tds_quad.hip / tds_oct.hip (see DevModel::quad, ::oct): root chain in closed form + four legs.  Both kernels
  // are built for the env step with PD control on the leg joints ONLY: the root dofs carry no torque, every action belongs to
  // a leg-joint lane (a TAU-mode model, a PD loop that starts inside the root chain or more actions than leg joints stay
  // on the general kernel, which handles them)
  auto star_actuation = [&](int leg_len, int leg_dofs) {
    bool ok = m->step_mode == TDS_STEP_LOCOMOTION && m->action_dim <= 4 * leg_dofs;
    for (int i = 0; ok && i < 6; ++i) ok = d->act_index[i] < 0;
    for (int i = 6; ok && i < m->num_links; ++i)
      ok = d->act_index[i] < 0 || ((i - 6) % leg_len < leg_dofs && d->act_index[i] < m->action_dim);
    return ok;
  };
  d->quad = 0;
  if (d->euler_root && sizeof(T) == 8 && m->num_links == 22 && m->dof_qd == 18 && m->dof_q == 18 && m->has_plane &&
      ncp == 4 && tds_opt_now(TDS_OPT_QUAD) != 0 && star_actuation(4, 3)) {
    bool ok = true;
    for (int k = 0; ok && k < 4; ++k) {
      for (int j = 0; ok && j < 4; ++j) {
        const int i = 6 + 4 * k + j;
        const tds_link_t &l = m->links[i];
        ok = l.parent == (j == 0 ? 5 : i - 1);
        if (j < 3)
          ok = ok && l.joint_type >= TDS_JOINT_PRISMATIC_X && l.joint_type <= TDS_JOINT_REVOLUTE_AXIS && l.qd_index == 6 + 3 * k + j &&
               l.q_index == 6 + 3 * k + j;
        else
          ok = ok && l.joint_type == TDS_JOINT_FIXED;
      }
      ok = ok && d->cp_link[k] == 9 + 4 * k;
    }
    ok = ok && (d->num_visuals == 0 || d->num_visuals == 17);
    for (int v = 0; ok && v < d->num_visuals; ++v) ok = d->vis_link[v] == 5 + v;
    d->quad = ok ? 1 : 0;
  }
  tds_oct_detect<T>(m, d, ncp, star_actuation(2, 2));
  tds_chain_detect<T>(m, d);
#undef TDS_FAIL
  return TDS_OK;
}

// ---- small rigid-transform / inertia helpers of the expansion (row-major 3x3, X = (R, t): child frame -> parent) ----
struct TdsXf { double R[9], t[3]; };
static inline TdsXf tds_xf_identity() { TdsXf x = {{1, 0, 0, 0, 1, 0, 0, 0, 1}, {0, 0, 0}}; return x; }
static inline TdsXf tds_xf_make(const double *R, const double *t) { TdsXf x; memcpy(x.R, R, sizeof(x.R)); memcpy(x.t, t, sizeof(x.t)); return x; }
static inline TdsXf tds_xf_mul(const TdsXf &a, const TdsXf &b) {  // (A*B).R = A.R B.R, .t = A.t + A.R B.t  (transform.hpp:123-131)
  TdsXf o;
  for (int r = 0; r < 3; ++r) {
    for (int c = 0; c < 3; ++c) o.R[3 * r + c] = a.R[3 * r] * b.R[c] + a.R[3 * r + 1] * b.R[3 + c] + a.R[3 * r + 2] * b.R[6 + c];
    o.t[r] = a.t[r] + a.R[3 * r] * b.t[0] + a.R[3 * r + 1] * b.t[1] + a.R[3 * r + 2] * b.t[2];
  }
  return o;
}
// rigid body (mass mb, com cb, inertia-about-com Ib, all in frame B) welded to link L at X (B in L's frame)
static inline void tds_weld_inertia(tds_link_t &L, const TdsXf &X, double mb, const double *cb, const double *Ib) {
  if (mb == 0.0) {
    bool zero = true;
    for (int k = 0; k < 9; ++k) zero &= Ib[k] == 0.0;
    if (zero) return;
  }
  double c2[3], RI[9], I2[9];
  for (int r = 0; r < 3; ++r) c2[r] = X.t[r] + X.R[3 * r] * cb[0] + X.R[3 * r + 1] * cb[1] + X.R[3 * r + 2] * cb[2];
  for (int r = 0; r < 3; ++r)
    for (int c = 0; c < 3; ++c) RI[3 * r + c] = X.R[3 * r] * Ib[c] + X.R[3 * r + 1] * Ib[3 + c] + X.R[3 * r + 2] * Ib[6 + c];
  for (int r = 0; r < 3; ++r)
    for (int c = 0; c < 3; ++c) I2[3 * r + c] = RI[3 * r] * X.R[3 * c] + RI[3 * r + 1] * X.R[3 * c + 1] + RI[3 * r + 2] * X.R[3 * c + 2];
  const double ma = L.mass, M = ma + mb;
  double c[3];
  for (int k = 0; k < 3; ++k) c[k] = M != 0.0 ? (ma * L.com[k] + mb * c2[k]) / M : 0.0;
  double I[9];
  for (int k = 0; k < 9; ++k) I[k] = L.inertia[k] + I2[k];
  const double da[3] = {L.com[0] - c[0], L.com[1] - c[1], L.com[2] - c[2]}, db[3] = {c2[0] - c[0], c2[1] - c[1], c2[2] - c[2]};
  const double da2 = da[0] * da[0] + da[1] * da[1] + da[2] * da[2], db2 = db[0] * db[0] + db[1] * db[1] + db[2] * db[2];
  for (int r = 0; r < 3; ++r)
    for (int cc = 0; cc < 3; ++cc)
      I[3 * r + cc] += ma * ((r == cc ? da2 : 0.0) - da[r] * da[cc]) + mb * ((r == cc ? db2 : 0.0) - db[r] * db[cc]);
  L.mass = M;
  memcpy(L.com, c, sizeof(c));
  memcpy(L.inertia, I, sizeof(I));
}

// Floating base, spherical joints, more links than lanes -> the expanded model the builder above understands:
//   * floating base: six pseudo links in front (dof nj+k, unit axes, link 5 carries mb.base_rbi()), the model's
//     links behind them with the base as link 5, joint dofs renumbered 0..nj-1;
//   * spherical joint: three lanes SPH0 / SPH1 / SPH2 (the last one is the link: inertia, shapes, children);
//   * JOINT_FIXED links below a moving link are FOLDED into it when the model would not fit the 32 lanes otherwise
//     (or with TDS_HIP_FOLD_FIXED=1): inertia welded on, shapes / visuals / children re-based through the constant
//     transform.  The reference keeps them as links with 1/D = 0 (forward_dynamics.hpp:153), which is the same
//     rigid body.
// Returns a malloc'ed TdsExpanded or NULL (why filled).
static inline TdsExpanded *tds_expand_model(const tds_model_t *m, char *why) {
  why[0] = 0;
#define TDS_XFAIL(msg)       \
  do {                       \
    strncpy(why, msg, 127);  \
    why[127] = 0;            \
    free(e);                 \
    return nullptr;          \
  } while (0)
  TdsExpanded *e = nullptr;
  if (m->num_links < 0 || m->num_links > TDS_MAX_LINKS) TDS_XFAIL("num_links out of range");
  const bool fl = m->is_floating != 0;
  int nsph = 0, nfixed_foldable = 0;
  bool moving_above[TDS_MAX_LINKS];  // some ancestor (or the floating base) moves: a fixed link here can be folded
  for (int i = 0; i < m->num_links; ++i) {
    const tds_link_t &l = m->links[i];
    if (l.parent >= i || l.parent < -1) TDS_XFAIL("links must be ordered parent-before-child");
    nsph += l.joint_type == TDS_JOINT_SPHERICAL;
    const bool par_moving = l.parent < 0 ? fl : (moving_above[l.parent] || m->links[l.parent].joint_type != TDS_JOINT_FIXED);
    moving_above[i] = par_moving;
    nfixed_foldable += (l.joint_type == TDS_JOINT_FIXED && par_moving) ? 1 : 0;
  }
  const int base = fl ? 6 : 0;
  if (fl && nsph) TDS_XFAIL("floating base + spherical joints: the reference fills the base/joint block of M one way only (mass_matrix.hpp:80-84); not built");
  const bool fold = (m->num_links + base + 2 * nsph > TDS_NL) || tds_opt_now_flag(TDS_OPT_FOLD_FIXED);
  if (m->num_links + base + 2 * nsph - (fold ? nfixed_foldable : 0) > TDS_NL)
    TDS_XFAIL("too many links for 32 lanes (one per moving link, 6 for a floating base, 3 per spherical joint)");
  const int nj = m->dof_qd - base;
  if (nj < 0 || m->dof_q != m->dof_qd + (fl ? 1 : 0) + nsph) TDS_XFAIL("dof_q / dof_qd inconsistent with the base and the joints");
  e = (TdsExpanded *)malloc(sizeof(TdsExpanded));
  if (!e) return nullptr;
  memcpy(&e->m, m, sizeof(*m));
  e->num_spherical = nsph;
  e->multi = 0;
  for (int k = 0; k < TDS_NB; ++k) e->body_fl[k] = 0;
  for (int k = 0; k < TDS_NL; ++k) {
    e->q_rec[k] = e->qd_rec[k] = -1;
    e->pd_on[k] = 0;
    e->sph_q[k] = e->fb_k[k] = e->fb_q[k] = e->tau_rec[k] = -1;
  }
  for (int k = 0; k < base; ++k) {
    tds_link_t &L = e->m.links[k];
    memset(&L, 0, sizeof(L));
    L.joint_type = k < 3 ? TDS_JOINT_REVOLUTE_X + k : TDS_JOINT_PRISMATIC_X + (k - 3);
    L.parent = k - 1;
    L.q_index = L.qd_index = nj + k;
    L.X_T_rot[0] = L.X_T_rot[4] = L.X_T_rot[8] = 1.0;
    L.S[k] = 1.0;
    e->qd_rec[k] = k;
  }
  if (fl) {
    e->m.links[5].mass = m->base_mass;
    memcpy(e->m.links[5].com, m->base_com, sizeof(m->base_com));
    memcpy(e->m.links[5].inertia, m->base_inertia, sizeof(m->base_inertia));
  }
  int carrier[TDS_MAX_LINKS];   // expanded index of the lane that carries link i (inertia, shapes, children); -1: base
  TdsXf xacc[TDS_MAX_LINKS];    // frame of link i in its carrier's frame (identity unless folded)
  int nx = base, ndof = 0, nq_rec = fl ? 7 : 0, nqd_rec = base;
  for (int i = 0; i < m->num_links; ++i) {
    const tds_link_t &l = m->links[i];
    const int par = l.parent < 0 ? (fl ? 5 : -1) : carrier[l.parent];
    const TdsXf xpar = l.parent < 0 ? tds_xf_identity() : xacc[l.parent];   // parent link's frame in ITS carrier
    const TdsXf xt = tds_xf_mul(xpar, tds_xf_make(l.X_T_rot, l.X_T_trans));  // this link's joint frame in that carrier
    xacc[i] = tds_xf_identity();
    if (l.joint_type == TDS_JOINT_FIXED && fold && moving_above[i]) {
      carrier[i] = par;  // (>= 0: moving_above)
      xacc[i] = xt;
      tds_weld_inertia(e->m.links[par], xt, l.mass, l.com, l.inertia);
      continue;
    }
    if (nx + (l.joint_type == TDS_JOINT_SPHERICAL ? 3 : 1) > TDS_NL) TDS_XFAIL("too many links for 32 lanes");
    const bool pd = m->step_mode == TDS_STEP_LOCOMOTION && i >= m->pd_start_link;
    if (l.joint_type == TDS_JOINT_SPHERICAL) {
      if (l.q_index != nq_rec || l.qd_index != nqd_rec) TDS_XFAIL("q/qd indices must be dense in link order");
      for (int k = 0; k < 3; ++k) {
        tds_link_t &L = e->m.links[nx];
        memset(&L, 0, sizeof(L));
        L.joint_type = TDS_JOINT_SPH0 + k;
        L.parent = k == 0 ? par : nx - 1;
        L.q_index = L.qd_index = ndof++;
        L.S[k] = 1.0;
        L.damping = l.damping;
        L.stiffness = l.stiffness;  // axis-angle spring (forward_dynamics.hpp:70-74): component k on lane k
        L.X_T_rot[0] = L.X_T_rot[4] = L.X_T_rot[8] = 1.0;
        if (k == 0) {
          memcpy(L.X_T_rot, xt.R, sizeof(L.X_T_rot));
          memcpy(L.X_T_trans, xt.t, sizeof(L.X_T_trans));
          e->q_rec[nx] = nq_rec;
        }
        if (k == 2) {
          L.mass = l.mass;
          memcpy(L.com, l.com, sizeof(L.com));
          memcpy(L.inertia, l.inertia, sizeof(L.inertia));
        }
        e->qd_rec[nx] = nqd_rec + k;
        e->pd_on[nx] = pd ? ((fl || i >= 4) ? 2 : 1) : 0;
        e->sph_q[nx] = nq_rec;
        ++nx;
      }
      carrier[i] = nx - 1;
      nq_rec += 4;
      nqd_rec += 3;
    } else {
      tds_link_t &L = e->m.links[nx];
      L = l;
      L.parent = par;
      memcpy(L.X_T_rot, xt.R, sizeof(L.X_T_rot));
      memcpy(L.X_T_trans, xt.t, sizeof(L.X_T_trans));
      if (l.joint_type != TDS_JOINT_FIXED) {
        // (the reference numbers q from 7 and qd from 6 on a floating base, multi_body.hpp:324-349)
        if (l.q_index != nq_rec || l.qd_index != nqd_rec) TDS_XFAIL("q/qd indices must be dense in link order");
        L.q_index = L.qd_index = ndof++;
        e->q_rec[nx] = nq_rec++;
        e->qd_rec[nx] = nqd_rec++;
      }
      e->pd_on[nx] = pd;
      carrier[i] = nx++;
    }
  }
  if (ndof != nj) TDS_XFAIL("dof_qd does not match the joints");
  e->m.num_links = nx;
  for (int g = 0; g < m->num_geoms && g < TDS_MAX_GEOMS; ++g) {
    const int lk = m->geoms[g].link;
    if (lk < -1 || lk >= m->num_links) TDS_XFAIL("geom link out of range");
    tds_geom_t &G = e->m.geoms[g];
    G.link = lk < 0 ? (fl ? 5 : -1) : carrier[lk];
    if (lk >= 0) {
      const TdsXf x = tds_xf_mul(xacc[lk], tds_xf_make(G.X_rot, G.X_trans));
      memcpy(G.X_rot, x.R, sizeof(G.X_rot));
      memcpy(G.X_trans, x.t, sizeof(G.X_trans));
    }
  }
  for (int v = 0; v < m->num_visuals && v < TDS_MAX_VISUALS; ++v) {
    const int lk = m->visuals[v].link;
    if (lk < 0 || lk >= m->num_links) TDS_XFAIL("visual link out of range");
    tds_visual_t &V = e->m.visuals[v];
    V.link = carrier[lk];
    const TdsXf x = tds_xf_mul(xacc[lk], tds_xf_make(V.X_rot, V.X_trans));
    memcpy(V.X_rot, x.R, sizeof(V.X_rot));
    memcpy(V.X_trans, x.t, sizeof(V.X_trans));
  }
  e->m.pd_start_link = 0;  // (superseded by pd_on)
#undef TDS_XFAIL
  return e;
}

// A world of several articulated bodies with floating bases among them -> the expanded model: body by body, the six
// pseudo links of a floating base (see tds_expand_model) in front of the body's links; internal dofs numbered per body
// joints first, base last, the bodies one behind the other (a block-diagonal joint-space inertia whose blocks each end
// in their base's 6 x 6).  Fixed links are kept as links, spherical joints are not taken.
// Returns a malloc'ed TdsExpanded or NULL (why filled).
static inline TdsExpanded *tds_expand_multibody(const tds_model_t *m, char *why) {
  why[0] = 0;
#define TDS_XFAIL(msg)       \
  do {                       \
    strncpy(why, msg, 127);  \
    why[127] = 0;            \
    free(e);                 \
    return nullptr;          \
  } while (0)
  TdsExpanded *e = nullptr;
  const int B = m->num_bodies;
  if (B < 2 || B > TDS_NB) TDS_XFAIL("num_bodies out of range");
  if (m->num_links < 0 || m->num_links > TDS_MAX_LINKS) TDS_XFAIL("num_links out of range");
  e = (TdsExpanded *)malloc(sizeof(TdsExpanded));
  if (!e) return nullptr;
  memcpy(&e->m, m, sizeof(*m));
  e->num_spherical = 0;
  e->multi = 1;
  for (int k = 0; k < TDS_NL; ++k) {
    e->q_rec[k] = e->qd_rec[k] = -1;
    e->pd_on[k] = 0;
    e->sph_q[k] = e->fb_k[k] = e->fb_q[k] = e->tau_rec[k] = -1;
  }
  int carrier[TDS_MAX_LINKS], base_lane[TDS_NB];
  int nx = 0, ndof = 0, nq_rec = 0, nqd_rec = 0, ntau = 0;
  for (int b = 0; b < B; ++b) {
    const int l0 = b == 0 ? 0 : m->bodies[b].first_link, l1 = b + 1 < B ? m->bodies[b + 1].first_link : m->num_links;
    if (l0 < 0 || l1 < l0 || l1 > m->num_links) TDS_XFAIL("bodies[].first_link out of range or not ascending");
    const bool fl = (b == 0 ? m->is_floating : m->bodies[b].is_floating) != 0;
    e->body_fl[b] = fl ? 1 : 0;
    int nj = 0;
    for (int i = l0; i < l1; ++i) {
      if (m->links[i].joint_type == TDS_JOINT_SPHERICAL) TDS_XFAIL("multi-body worlds: 1-dof joints");
      nj += m->links[i].joint_type != TDS_JOINT_FIXED;
    }
    if (nx + (fl ? 6 : 0) + (l1 - l0) > TDS_NL) TDS_XFAIL("too many links for 32 lanes (one per link, 6 per floating base)");
    if (!fl && l1 == l0) TDS_XFAIL("a body with a fixed base and no links");
    if (b > 0) e->m.bodies[b].first_link = nx;  // (expanded index)
    base_lane[b] = fl ? nx + 5 : -1;
    if (fl) {
      const int bd = ndof + nj;  // the base's six dofs follow the body's joints
      for (int k = 0; k < 6; ++k) {
        tds_link_t &L = e->m.links[nx];
        memset(&L, 0, sizeof(L));
        L.joint_type = k < 3 ? TDS_JOINT_REVOLUTE_X + k : TDS_JOINT_PRISMATIC_X + (k - 3);
        L.parent = k == 0 ? -1 : nx - 1;
        L.q_index = L.qd_index = bd + k;
        L.X_T_rot[0] = L.X_T_rot[4] = L.X_T_rot[8] = 1.0;
        L.S[k] = 1.0;
        e->qd_rec[nx] = nqd_rec + k;
        e->fb_k[nx] = k;
        e->fb_q[nx] = nq_rec;
        ++nx;
      }
      tds_link_t &L5 = e->m.links[nx - 1];
      L5.mass = b == 0 ? m->base_mass : m->bodies[b].base_mass;
      memcpy(L5.com, b == 0 ? m->base_com : m->bodies[b].base_com, sizeof(L5.com));
      memcpy(L5.inertia, b == 0 ? m->base_inertia : m->bodies[b].base_inertia, sizeof(L5.inertia));
      nq_rec += 7;   // [quat xyzw | pos], then the joints (multi_body.hpp:324-349)
      nqd_rec += 6;  // [omega | v]
    }
    for (int i = l0; i < l1; ++i) {
      const tds_link_t &l = m->links[i];
      if (l.parent >= i || l.parent < -1 || (l.parent >= 0 && l.parent < l0))
        TDS_XFAIL("links must be ordered parent-before-child within their body");
      tds_link_t &L = e->m.links[nx];
      L = l;
      L.parent = l.parent < 0 ? base_lane[b] : carrier[l.parent];
      if (l.joint_type != TDS_JOINT_FIXED) {
        if (l.q_index != nq_rec || l.qd_index != nqd_rec) TDS_XFAIL("q/qd indices must be dense in link order over all bodies");
        L.q_index = L.qd_index = ndof++;
        e->q_rec[nx] = nq_rec++;
        e->qd_rec[nx] = nqd_rec++;
        e->tau_rec[nx] = ntau++;
      }
      carrier[i] = nx++;
    }
    if (fl) ndof += 6;
  }
  if (ndof != m->dof_qd || nq_rec != m->dof_q || nqd_rec != m->dof_qd) TDS_XFAIL("dof_q / dof_qd inconsistent with the bases and the joints");
  e->m.num_links = nx;
  e->m.is_floating = 0;  // (per body: body_fl)
  for (int g = 0; g < m->num_geoms && g < TDS_MAX_GEOMS; ++g) {
    const int lk = m->geoms[g].link;
    if (lk < -B || lk >= m->num_links) TDS_XFAIL("geom link out of range");
    e->m.geoms[g].link = lk < 0 ? (base_lane[-1 - lk] >= 0 ? base_lane[-1 - lk] : lk) : carrier[lk];
  }
  for (int v = 0; v < m->num_visuals && v < TDS_MAX_VISUALS; ++v) {
    const int lk = m->visuals[v].link;
    if (lk < 0 || lk >= m->num_links) TDS_XFAIL("visual link out of range");
    e->m.visuals[v].link = carrier[lk];
  }
  e->m.pd_start_link = 0;
#undef TDS_XFAIL
  return e;
}

// Returns TDS_OK or an error code; `why` (>= 128 bytes) receives the reason.
template <typename T>
static int tds_build_dev_model(const tds_model_t *m, DevModel<T> *d, char *why) {
  if (m->abi_version != TDS_HIP_ABI_VERSION) {
    strncpy(why, "model abi_version mismatch", 127);
    return TDS_ERR_INVALID_ARG;
  }
  if (m->num_bodies >= 2 && m->num_bodies <= TDS_NB) {
    bool any_fl = m->is_floating != 0;
    for (int b = 1; b < m->num_bodies; ++b) any_fl |= m->bodies[b].is_floating != 0;
    if (any_fl) {
      TdsExpanded *e = tds_expand_multibody(m, why);
      if (!e) return TDS_ERR_UNSUPPORTED;
      const int rc = tds_build_dev_model_impl<T>(&e->m, d, why, e);
      free(e);
      return rc;
    }
  }
  bool general = m->is_floating != 0 || m->num_links > TDS_NL || tds_opt_now_flag(TDS_OPT_FOLD_FIXED);
  for (int i = 0; i < m->num_links && i < TDS_MAX_LINKS; ++i) general |= m->links[i].joint_type == TDS_JOINT_SPHERICAL;
  if (!general) return tds_build_dev_model_impl<T>(m, d, why, nullptr);
  TdsExpanded *e = tds_expand_model(m, why);
  if (!e) return TDS_ERR_UNSUPPORTED;
  const int rc = tds_build_dev_model_impl<T>(&e->m, d, why, e);
  free(e);
  return rc;
}
