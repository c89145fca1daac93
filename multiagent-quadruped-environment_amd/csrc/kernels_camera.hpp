// kernels_camera.hpp -- k_depth_camera: the onboard forward depth camera of LeggedRobotField (reference legged_robot_field.py:23-93,
// 196-223: gym.create_camera_sensor + attach_camera_to_body(FOLLOW_TRANSFORM) + get_camera_image_gpu_tensor(IMAGE_DEPTH)) as a ray
// caster over the geometry the physics collides with.  The reference renders with Isaac Gym's rasteriser, which is closed and absent:
// there is no image to compare with, so the contract is geometric -- for every pixel the distance ALONG THE OPTICAL AXIS to the first
// surface of: the ground (slab plane or relief map), the wall prisms (signed-distance map + wall tops), the OTHER robots' collision
// primitives (the URDF's boxes, capsules, spheres), the free NPCs (spheres / the box), the 1-dof link's plank and platform, the scenery
// boxes -- reported as Isaac Gym's IMAGE_DEPTH does: NEGATIVE depth in metres, -inf where nothing is hit within `far`.
// Camera frame = the body-local transform (position, ZYX Euler) of cfg.sensor.forward_camera on the base link, looking along its +x with
// +z up (Isaac Gym's attach convention); pixel (0, 0) is the top-left corner, the horizontal field of view is given, the vertical one
// follows from the aspect ratio.  Known answers: tests/test_camera_gpu.py.
// One 256-thread workgroup per env: the link frames of its robots by forward kinematics into LDS, the primitives in world coordinates
// behind them, then thread = pixel, looping over the env's cameras.
#pragma once
#include "mqe_common.hpp"

struct CamArgs { float* out; int H, W; float tan_half_h; float pos[3]; float rpy[3]; float far_; };

struct CV3 { float x, y, z; };
__device__ __forceinline__ CV3 cv(float x, float y, float z) { CV3 r; r.x = x; r.y = y; r.z = z; return r; }
__device__ __forceinline__ CV3 operator+(CV3 a, CV3 b) { return cv(a.x + b.x, a.y + b.y, a.z + b.z); }
__device__ __forceinline__ CV3 operator-(CV3 a, CV3 b) { return cv(a.x - b.x, a.y - b.y, a.z - b.z); }
__device__ __forceinline__ CV3 operator*(float s, CV3 a) { return cv(s * a.x, s * a.y, s * a.z); }
__device__ __forceinline__ float cdot(CV3 a, CV3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
__device__ __forceinline__ CV3 cmul(const float* R, CV3 v) {      // R v, row-major
  return cv(R[0] * v.x + R[1] * v.y + R[2] * v.z, R[3] * v.x + R[4] * v.y + R[5] * v.z, R[6] * v.x + R[7] * v.y + R[8] * v.z);
}
__device__ __forceinline__ CV3 cmulT(const float* R, CV3 v) {     // R^T v
  return cv(R[0] * v.x + R[3] * v.y + R[6] * v.z, R[1] * v.x + R[4] * v.y + R[7] * v.z, R[2] * v.x + R[5] * v.y + R[8] * v.z);
}
__device__ __forceinline__ void cmat(const float* A, const float* B, float* C) {
  for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) C[r * 3 + c] = A[r * 3] * B[c] + A[r * 3 + 1] * B[3 + c] + A[r * 3 + 2] * B[6 + c];
}
__device__ __forceinline__ void cquat(const float* q, float* R) {    // (x, y, z, w), normalised here
  float x = q[0], y = q[1], z = q[2], w = q[3];
  const float in = 1.0f / sqrtf(x * x + y * y + z * z + w * w);
  x *= in; y *= in; z *= in; w *= in;
  R[0] = 1 - 2 * (y * y + z * z); R[1] = 2 * (x * y - z * w); R[2] = 2 * (x * z + y * w);
  R[3] = 2 * (x * y + z * w); R[4] = 1 - 2 * (x * x + z * z); R[5] = 2 * (y * z - x * w);
  R[6] = 2 * (x * z - y * w); R[7] = 2 * (y * z + x * w); R[8] = 1 - 2 * (x * x + y * y);
}
__device__ __forceinline__ void caxis(CV3 a, float q, float* R) {    // rotation by q about the unit axis a
  const float c = cosf(q), s = sinf(q), t = 1 - c;
  R[0] = t * a.x * a.x + c; R[1] = t * a.x * a.y - s * a.z; R[2] = t * a.x * a.z + s * a.y;
  R[3] = t * a.x * a.y + s * a.z; R[4] = t * a.y * a.y + c; R[5] = t * a.y * a.z - s * a.x;
  R[6] = t * a.x * a.z - s * a.y; R[7] = t * a.y * a.z + s * a.x; R[8] = t * a.z * a.z + c;
}
// ray o + t d (d NOT normalised: t is the depth along the optical axis), nearest t in (tmin, best) or best
__device__ __forceinline__ float ray_sphere(CV3 o, CV3 d, CV3 c, float r, float best) {
  const CV3 oc = o - c;
  const float a = cdot(d, d), b = cdot(d, oc), cc = cdot(oc, oc) - r * r;
  const float h = b * b - a * cc;
  if (h < 0.0f || cc < 0.0f) return best;                     // miss, or the eye is inside
  const float t = (-b - sqrtf(h)) / a;
  return (t > 1e-4f && t < best) ? t : best;
}
__device__ __forceinline__ float ray_capsule(CV3 o, CV3 d, CV3 c, CV3 u, float r, float best) {
  const CV3 pa = c - u, ba = 2.0f * u, oa = o - pa;
  const float baba = cdot(ba, ba), bard = cdot(ba, d), baoa = cdot(ba, oa), rdoa = cdot(d, oa), oaoa = cdot(oa, oa), dd = cdot(d, d);
  if (baba < 1e-12f) return ray_sphere(o, d, c, r, best);
  const float a = baba * dd - bard * bard, b = baba * rdoa - baoa * bard, cc = baba * oaoa - baoa * baoa - r * r * baba;
  const float h = b * b - a * cc;
  if (h >= 0.0f && a > 1e-12f) {
    const float t = (-b - sqrtf(h)) / a;
    const float y = baoa + t * bard;
    if (y > 0.0f && y < baba) return (t > 1e-4f && t < best) ? t : best;
  }
  float t1 = ray_sphere(o, d, pa, r, best);
  return ray_sphere(o, d, c + u, r, t1);
}
__device__ __forceinline__ float ray_box(CV3 o, CV3 d, CV3 c, const float* R, CV3 h, float best) {      // oriented box: slabs in its frame
  const CV3 ol = cmulT(R, o - c), dl = cmulT(R, d);
  float t0 = 1e-4f, t1 = best;
  const float oo[3] = {ol.x, ol.y, ol.z}, dv[3] = {dl.x, dl.y, dl.z}, hh[3] = {h.x, h.y, h.z};
#pragma unroll
  for (int k = 0; k < 3; k++) {
    if (fabsf(dv[k]) < 1e-9f) { if (fabsf(oo[k]) > hh[k]) return best; continue; }
    const float inv = 1.0f / dv[k];
    float ta = (-hh[k] - oo[k]) * inv, tb = (hh[k] - oo[k]) * inv;
    if (ta > tb) { const float s = ta; ta = tb; tb = s; }
    t0 = fmaxf(t0, ta); t1 = fminf(t1, tb);
    if (t0 > t1) return best;
  }
  return (fabsf(oo[0]) <= hh[0] && fabsf(oo[1]) <= hh[1] && fabsf(oo[2]) <= hh[2]) ? best : t0;      // (an eye inside the box sees nothing of it)
}

#define CAM_LINK 12          // floats per link frame in LDS: R (9), origin (3)
#define CAM_PRIM 16          // floats per primitive: type, link, radius, -, centre (3), -, half-axis / half extents (3), -, (4 spare)
__global__ void __launch_bounds__(256) k_depth_camera(const DevModel* __restrict__ m, DevState st, CamArgs ca) {
  __shared__ float s_link[MQE_MAX_AGENTS * MQE_NBODY * CAM_LINK];
  __shared__ float s_prim[MQE_MAX_AGENTS * MQE_MAX_PRIMS * CAM_PRIM];
  const int e = blockIdx.x, tid = threadIdx.x;
  const int A = m->A, P = m->P, npr = m->robot.n_prims;
  const mqe_robot_model& rm = m->robot;
  const float* root = st.root + (size_t)e * (A + P) * 13;
  const float* dof = st.dof + (size_t)e * m->ND * 2;
  // ---- link frames: thread = (robot, body), walking its chain from the base (<= 3 joints) -------------------------------------------
  if (tid < A * MQE_NBODY) {
    const int r = tid / MQE_NBODY, bb = tid - r * MQE_NBODY;
    float R[9];
    cquat(root + r * 13 + 3, R);
    CV3 p = cv(root[r * 13], root[r * 13 + 1], root[r * 13 + 2]);
    if (bb > 0) {
      const int leg = (bb - 1) / 3, t = (bb - 1) - leg * 3;
      for (int k = 0; k <= t; k++) {
        const int b = 1 + leg * 3 + k;
        p = p + cmul(R, cv(rm.joint_offset[b][0], rm.joint_offset[b][1], rm.joint_offset[b][2]));
        float Rj[9], Rn[9];
        caxis(cv(rm.joint_axis[b][0], rm.joint_axis[b][1], rm.joint_axis[b][2]), dof[(r * 12 + b - 1) * 2], Rj);
        cmat(R, Rj, Rn);
        for (int i = 0; i < 9; i++) R[i] = Rn[i];
      }
    }
    float* L = s_link + tid * CAM_LINK;
    for (int i = 0; i < 9; i++) L[i] = R[i];
    L[9] = p.x; L[10] = p.y; L[11] = p.z;
  }
  __syncthreads();
  // ---- the robots' primitives in world coordinates ----------------------------------------------------------------------------------
  if (tid < A * npr) {
    const int r = tid / npr, q = tid - r * npr;
    const int link = r * MQE_NBODY + rm.prim_body[q];
    const float* L = s_link + link * CAM_LINK;
    const CV3 c = cv(L[9], L[10], L[11]) + cmul(L, cv(rm.prim_center[q][0], rm.prim_center[q][1], rm.prim_center[q][2]));
    float* Q = s_prim + tid * CAM_PRIM;
    Q[0] = (float)rm.prim_type[q]; Q[1] = (float)link; Q[2] = rm.prim_half[q][0];
    Q[4] = c.x; Q[5] = c.y; Q[6] = c.z;
    if (rm.prim_type[q] == MQE_PRIM_BOX) { Q[8] = rm.prim_half[q][0]; Q[9] = rm.prim_half[q][1]; Q[10] = rm.prim_half[q][2]; }
    else { const CV3 u = cmul(L, cv(rm.prim_axis[q][0], rm.prim_axis[q][1], rm.prim_axis[q][2])); Q[8] = u.x; Q[9] = u.y; Q[10] = u.z; }
  }
  __syncthreads();
  // ---- rays --------------------------------------------------------------------------------------------------------------------------
  float Rc[9];                                                   // camera in the base frame: ZYX Euler (yaw, pitch, roll) = Rz Ry Rx
  {
    float Rz[9], Ry[9], Rx[9], T[9];
    caxis(cv(0, 0, 1), ca.rpy[2], Rz); caxis(cv(0, 1, 0), ca.rpy[1], Ry); caxis(cv(1, 0, 0), ca.rpy[0], Rx);
    cmat(Rz, Ry, T); cmat(T, Rx, Rc);
  }
  const int npix = ca.H * ca.W;
  const float tan_v = ca.tan_half_h * (float)ca.H / (float)ca.W;
  const float hs = m->hs;
  const int nx = m->sdf_nx, ny = m->sdf_ny;
  for (int idx = tid; idx < A * npix; idx += blockDim.x) {
    const int a = idx / npix, pix = idx - a * npix, pi = pix / ca.W, pj = pix - pi * ca.W;
    const float* B = s_link + (a * MQE_NBODY) * CAM_LINK;          // the base link of robot a
    float Rw[9];
    cmat(B, Rc, Rw);
    const CV3 o = cv(B[9], B[10], B[11]) + cmul(B, cv(ca.pos[0], ca.pos[1], ca.pos[2]));
    const float yc = -(2.0f * (pj + 0.5f) / ca.W - 1.0f) * ca.tan_half_h;      // column 0 = the camera's left (+y)
    const float zc = -(2.0f * (pi + 0.5f) / ca.H - 1.0f) * tan_v;               // row 0 = the top (+z)
    const CV3 d = cmul(Rw, cv(1.0f, yc, zc));                                    // t = depth along the optical axis
    float best = ca.far_;
    // ground: the slab plane, or the relief map marched in half cells and bisected
    if (m->ground_height == nullptr) {
      if (d.z < -1e-9f) { const float t = (m->ground_z - o.z) / d.z; if (t > 1e-4f && t < best) best = t; }
    } else {
      const float dxy = sqrtf(d.x * d.x + d.y * d.y);
      const float dt = dxy > 1e-6f ? 0.5f * hs / dxy : 0.05f;
      auto hgt = [&](CV3 p) -> float {
        float fx = fminf(fmaxf(p.x / hs, 0.0f), (float)(nx - 1)), fy = fminf(fmaxf(p.y / hs, 0.0f), (float)(ny - 1));
        int ix = min((int)fx, nx - 2), iy = min((int)fy, ny - 2);
        const float tx = fx - ix, ty = fy - iy;
        const float* g = m->ground_height + (size_t)ix * ny + iy;
        const float b0 = g[0] + (g[1] - g[0]) * ty, b1 = g[ny] + (g[ny + 1] - g[ny]) * ty;
        return m->ground_z + b0 + (b1 - b0) * tx;
      };
      float tp = 0.0f;
      for (float t = dt; t < best; t += dt) {
        const CV3 p = o + t * d;
        if (p.z < hgt(p)) {
          float lo = tp, hi = t;
          for (int it = 0; it < 6; it++) { const float mid = 0.5f * (lo + hi); const CV3 pm = o + mid * d; if (pm.z < hgt(pm)) hi = mid; else lo = mid; }
          best = hi;
          break;
        }
        tp = t;
      }
    }
    // wall prisms: march over the signed-distance map of the wall set (a step never crosses more than the distance to the nearest wall)
    {
      const float dxy = sqrtf(d.x * d.x + d.y * d.y);
      if (dxy > 1e-6f && m->wall_sdf != nullptr) {
        float t = 1e-4f;
        for (int it = 0; it < 400 && t < best; it++) {
          const CV3 p = o + t * d;
          const float fxr = p.x / hs, fyr = p.y / hs;
          if (fxr < 0.0f || fyr < 0.0f || fxr > (float)(nx - 1) || fyr > (float)(ny - 1)) break;       // left the map
          int ix = min((int)fxr, nx - 2), iy = min((int)fyr, ny - 2);
          const float tx = fxr - ix, ty = fyr - iy;
          const float* sd = m->wall_sdf + (size_t)ix * ny + iy;
          const float a0 = sd[0] + (sd[1] - sd[0]) * ty, a1 = sd[ny] + (sd[ny + 1] - sd[ny]) * ty;
          const float s = a0 + (a1 - a0) * tx;
          if (s <= 0.002f) {                                     // on / inside a wall's footprint: below its top it is the wall
            const float top = m->wall_top != nullptr ? m->wall_top[(size_t)(tx < 0.5f ? ix : ix + 1) * ny + (ty < 0.5f ? iy : iy + 1)] : m->wall_height;
            if (p.z <= top && p.z >= m->ground_z - 1e-3f) { best = t; break; }
            t += 0.25f * hs / dxy;                               // above it: the ray may still come down onto the top face
          } else t += fmaxf(s, 0.002f) / dxy;
        }
      }
    }
    // the other robots' primitives
    for (int r = 0; r < A; r++) {
      if (r == a) continue;
      for (int q = 0; q < npr; q++) {
        const float* Q = s_prim + (r * npr + q) * CAM_PRIM;
        const int type = (int)Q[0];
        const CV3 c = cv(Q[4], Q[5], Q[6]);
        if (type == MQE_PRIM_BOX) best = ray_box(o, d, c, s_link + (int)Q[1] * CAM_LINK, cv(Q[8], Q[9], Q[10]), best);
        else if (type == MQE_PRIM_CAPSULE) best = ray_capsule(o, d, c, cv(Q[8], Q[9], Q[10]), Q[2], best);
        else best = ray_sphere(o, d, c, Q[2], best);
      }
    }
    // free NPCs (ball, sheep: spheres in the body frame, translation only for the sheep; the box), the 1-dof link, the scenery
    if (m->has_box) {
      for (int p = 0; p < m->n_npc_dyn; p++) {
        float Rb[9];
        cquat(root + (A + p) * 13 + 3, Rb);
        best = ray_box(o, d, cv(root[(A + p) * 13], root[(A + p) * 13 + 1], root[(A + p) * 13 + 2]), Rb, cv(m->npc_box_half[0], m->npc_box_half[1], m->npc_box_half[2]), best);
      }
    } else {
      for (int p = 0; p < m->n_npc_dyn; p++) {
        float Rb[9];
        cquat(root + (A + p) * 13 + 3, Rb);
        const CV3 pb = cv(root[(A + p) * 13], root[(A + p) * 13 + 1], root[(A + p) * 13 + 2]);
        for (int k = 0; k < m->npc_n_spheres; k++)
          best = ray_sphere(o, d, pb + cmul(Rb, cv(m->npc_sphere_center[k][0], m->npc_sphere_center[k][1], m->npc_sphere_center[k][2])), m->npc_sphere_radius[k], best);
      }
    }
    const float I3[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
    if (m->has_seesaw) {
      const CV3 sb = cv(root[A * 13], root[A * 13 + 1], root[A * 13 + 2]);
      if (m->ss_base_half[0] > 0.0f) best = ray_box(o, d, sb, I3, cv(m->ss_base_half[0], m->ss_base_half[1], m->ss_base_half[2]), best);
      CV3 piv = sb + cv(m->ss_joint_offset[0], m->ss_joint_offset[1], m->ss_joint_offset[2]);
      const float th = dof[(12 * A) * 2];
      float Rp[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
      if (m->ss_axis == 3) piv.y += th; else caxis(m->ss_axis == 2 ? cv(0, 0, 1) : cv(0, 1, 0), th, Rp);
      const CV3 pc = piv + cmul(Rp, cv(m->ss_plank_center[0], m->ss_plank_center[1], m->ss_plank_center[2]));
      if (m->ss_link_cyl) best = ray_capsule(o, d, pc, cv(0, 0, fmaxf(m->ss_plank_half[2] - m->ss_plank_half[0], 0.0f)), m->ss_plank_half[0], best);      // upright cylinder ~ capsule
      else best = ray_box(o, d, pc, Rp, cv(m->ss_plank_half[0], m->ss_plank_half[1], m->ss_plank_half[2]), best);
    }
    for (int bx = 0; bx < m->n_static; bx++) {
      const CV3 nb = cv(root[A * 13], root[A * 13 + 1], root[A * 13 + 2]);
      best = ray_box(o, d, nb + cv(m->sb_center[bx][0], m->sb_center[bx][1], m->sb_center[bx][2]), I3, cv(m->sb_half[bx][0], m->sb_half[bx][1], m->sb_half[bx][2]), best);
    }
    ca.out[((size_t)e * A + a) * npix + pix] = best < ca.far_ ? -best : __uint_as_float(0xFF800000u);       // -inf (the build drops inf arithmetic: written as its bit pattern)
  }
}
