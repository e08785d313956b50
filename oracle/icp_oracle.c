/* icp_oracle.c -- CPU restatement of the mp2p_icp ICP::align() hot path.  See icp_oracle.h:
 * TEST INFRASTRUCTURE ONLY; PARITY UNPINNED (no golden vectors exist for this path in the
 * reference; this restates SURVEY.md 8(a)+Appendix A, cross-checked by icp_oracle_np.py).
 *
 * Build: gcc -O3 -ffp-contract=off -fopenmp -shared -fPIC  (see oracle/Makefile).
 * -ffp-contract=off is REQUIRED: the fp32 distance arithmetic and the double->float point
 * transform are specified un-fused so that the HIP kernels can reproduce them bit for bit.
 */
#include "icp_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif


/* ======================================================================================
 * SE(3) helpers -- SURVEY Appendix A (mrpt::poses::CPose3D, Lie::SE<3>, Lie::SO<3>)
 * ==================================================================================== */
#define R_(T, i, j) ((T)[(i)*4 + (j)])
#define t_(T, i) ((T)[(i)*4 + 3])

void orc_pose_from_ypr(const double p[6], double T[12]) {
  /* R = Rz(yaw)*Ry(pitch)*Rx(roll): TPose3D ordering, LidarOdometry.cpp:235 */
  const double cy = cos(p[3]), sy = sin(p[3]);
  const double cp = cos(p[4]), sp = sin(p[4]);
  const double cr = cos(p[5]), sr = sin(p[5]);
  R_(T, 0, 0) = cy * cp; R_(T, 0, 1) = cy * sp * sr - sy * cr; R_(T, 0, 2) = cy * sp * cr + sy * sr;
  R_(T, 1, 0) = sy * cp; R_(T, 1, 1) = sy * sp * sr + cy * cr; R_(T, 1, 2) = sy * sp * cr - cy * sr;
  R_(T, 2, 0) = -sp;     R_(T, 2, 1) = cp * sr;                R_(T, 2, 2) = cp * cr;
  t_(T, 0) = p[0]; t_(T, 1) = p[1]; t_(T, 2) = p[2];
}

void orc_pose_to_ypr(const double T[12], double p[6]) {
  p[0] = t_(T, 0); p[1] = t_(T, 1); p[2] = t_(T, 2);
  const double c = hypot(R_(T, 0, 0), R_(T, 1, 0));
  p[4] = atan2(-R_(T, 2, 0), c);
  if (c > 1e-12) {
    p[3] = atan2(R_(T, 1, 0), R_(T, 0, 0));
    p[5] = atan2(R_(T, 2, 1), R_(T, 2, 2));
  } else { /* gimbal lock: put everything in yaw */
    p[3] = atan2(-R_(T, 0, 1), R_(T, 1, 1));
    p[5] = 0.0;
  }
}

void orc_pose_compose(const double A[12], const double B[12], double C[12]) {
  double out[12];
  for (int i = 0; i < 3; i++) {
    for (int j = 0; j < 3; j++)
      out[i * 4 + j] = R_(A, i, 0) * R_(B, 0, j) + R_(A, i, 1) * R_(B, 1, j) + R_(A, i, 2) * R_(B, 2, j);
    out[i * 4 + 3] = R_(A, i, 0) * t_(B, 0) + R_(A, i, 1) * t_(B, 1) + R_(A, i, 2) * t_(B, 2) + t_(A, i);
  }
  memcpy(C, out, sizeof(out));
}

void orc_pose_inverse(const double A[12], double B[12]) {
  double out[12];
  for (int i = 0; i < 3; i++) {
    for (int j = 0; j < 3; j++) out[i * 4 + j] = R_(A, j, i);
    out[i * 4 + 3] = -(R_(A, 0, i) * t_(A, 0) + R_(A, 1, i) * t_(A, 1) + R_(A, 2, i) * t_(A, 2));
  }
  memcpy(B, out, sizeof(out));
}

static void skew_sq_coeffs(double th, double* a, double* b, double* c) {
  /* a = sin(th)/th, b = (1-cos th)/th^2, c = (th - sin th)/th^3, cancellation-free */
  const double t2 = th * th;
  if (th < 1e-2) {
    *a = 1.0 - t2 / 6.0 * (1.0 - t2 / 20.0 * (1.0 - t2 / 42.0));
    *b = 0.5 - t2 / 24.0 * (1.0 - t2 / 30.0 * (1.0 - t2 / 56.0));
    *c = 1.0 / 6.0 - t2 / 120.0 * (1.0 - t2 / 42.0 * (1.0 - t2 / 72.0));
  } else {
    const double sh = sin(0.5 * th);
    *a = sin(th) / th;
    *b = 2.0 * sh * sh / t2;
    *c = (th - sin(th)) / (t2 * th);
  }
}

void orc_se3_exp(const double xi[6], double T[12]) {
  const double wx = xi[3], wy = xi[4], wz = xi[5];
  const double th = sqrt(wx * wx + wy * wy + wz * wz);
  double a, b, c;
  skew_sq_coeffs(th, &a, &b, &c);
  const double W[9] = {0, -wz, wy, wz, 0, -wx, -wy, wx, 0};
  double W2[9];
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) W2[i * 3 + j] = W[i * 3 + 0] * W[0 * 3 + j] + W[i * 3 + 1] * W[1 * 3 + j] + W[i * 3 + 2] * W[2 * 3 + j];
  double V[9];
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) {
      const double I = (i == j) ? 1.0 : 0.0;
      R_(T, i, j) = I + a * W[i * 3 + j] + b * W2[i * 3 + j];
      V[i * 3 + j] = I + b * W[i * 3 + j] + c * W2[i * 3 + j];
    }
  for (int i = 0; i < 3; i++) t_(T, i) = V[i * 3 + 0] * xi[0] + V[i * 3 + 1] * xi[1] + V[i * 3 + 2] * xi[2];
}

void orc_so3_log(const double T[12], double w[3]) {
  const double tr = R_(T, 0, 0) + R_(T, 1, 1) + R_(T, 2, 2);
  double cth = 0.5 * (tr - 1.0);
  if (cth > 1.0) cth = 1.0;
  if (cth < -1.0) cth = -1.0;
  const double vx = R_(T, 2, 1) - R_(T, 1, 2), vy = R_(T, 0, 2) - R_(T, 2, 0), vz = R_(T, 1, 0) - R_(T, 0, 1);
  const double s2 = sqrt(vx * vx + vy * vy + vz * vz); /* = 2 sin(th) */
  const double th = atan2(0.5 * s2, cth);
  if (th < 1e-7) {
    const double k = 0.5 * (1.0 + th * th / 6.0);
    w[0] = k * vx; w[1] = k * vy; w[2] = k * vz;
    return;
  }
  if (3.141592653589793 - th > 1e-6) {
    const double k = th / s2;
    w[0] = k * vx; w[1] = k * vy; w[2] = k * vz;
    return;
  }
  /* near pi: axis from the largest diagonal of (R + I)/2 = n n^T (+O(pi-th)) */
  double n[3];
  const double d0 = R_(T, 0, 0), d1 = R_(T, 1, 1), d2 = R_(T, 2, 2);
  int k = (d0 >= d1 && d0 >= d2) ? 0 : (d1 >= d2 ? 1 : 2);
  const double nk = sqrt(fmax(0.0, 0.5 * (R_(T, k, k) + 1.0)));
  n[k] = nk;
  for (int j = 0; j < 3; j++)
    if (j != k) n[j] = 0.25 * (R_(T, k, j) + R_(T, j, k)) / nk;
  /* fix sign with the antisymmetric part when it is informative */
  const double v[3] = {vx, vy, vz};
  const double dot = n[0] * v[0] + n[1] * v[1] + n[2] * v[2];
  const double sgn = (dot < 0.0) ? -1.0 : 1.0;
  const double nn = sqrt(n[0] * n[0] + n[1] * n[1] + n[2] * n[2]);
  for (int j = 0; j < 3; j++) w[j] = sgn * th * n[j] / nn;
}

void orc_se3_log(const double T[12], double xi[6]) {
  double w[3];
  orc_so3_log(T, w);
  const double th = sqrt(w[0] * w[0] + w[1] * w[1] + w[2] * w[2]);
  /* V^-1 = I - W/2 + k W^2, k = (1 - th sin th / (2(1-cos th)))/th^2 */
  double k;
  if (th < 1e-2)
    k = 1.0 / 12.0 + th * th / 720.0 + th * th * th * th / 30240.0;
  else
    k = (1.0 - 0.5 * th / tan(0.5 * th)) / (th * th);
  const double W[9] = {0, -w[2], w[1], w[2], 0, -w[0], -w[1], w[0], 0};
  double W2[9];
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) W2[i * 3 + j] = W[i * 3 + 0] * W[0 * 3 + j] + W[i * 3 + 1] * W[1 * 3 + j] + W[i * 3 + 2] * W[2 * 3 + j];
  for (int i = 0; i < 3; i++) {
    double acc = 0;
    for (int j = 0; j < 3; j++) {
      const double Vinv = ((i == j) ? 1.0 : 0.0) - 0.5 * W[i * 3 + j] + k * W2[i * 3 + j];
      acc += Vinv * t_(T, j);
    }
    xi[i] = acc;
  }
  xi[3] = w[0]; xi[4] = w[1]; xi[5] = w[2];
}

/* ======================================================================================
 * Local map -- mola::HashedVoxelPointCloud restated (yaml:228-242)
 * ==================================================================================== */
typedef struct {
  int32_t k[3];
  uint32_t n, cap;
  float* xyz;    /* n x 3 interleaved, insertion order */
  uint32_t* src; /* source index of each stored point */
  /* NDT statistics (mola::NDT role), recomputed lazily after insertions */
  float ndt_c[3], ndt_n[3];
  uint8_t ndt_plane, ndt_dirty;
} voxel_t;

struct orc_map {
  orc_map_params p;
  float inv_vs;
  voxel_t* vox;
  size_t n_vox, cap_vox;
  int32_t* table; /* open addressing: voxel id or -1 */
  size_t table_size; /* power of two */
  size_t n_points;
  uint64_t n_offered;
  float bb_min[3], bb_max[3];
};

static inline uint64_t hash3(int32_t x, int32_t y, int32_t z) {
  /* the classic 3-prime spatial hash; any hash gives identical results */
  uint64_t h = ((uint64_t)(uint32_t)x * 73856093u) ^ ((uint64_t)(uint32_t)y * 19349663u) ^ ((uint64_t)(uint32_t)z * 83492791u);
  h ^= h >> 15;
  return h;
}

static inline int32_t coord2idx(const orc_map* m, float c) {
  /* idx = floor(coord * (1/voxel_size)) (SURVEY Appendix A "Voxel indexing"); TRUNC is the
   * alternative reading kept as a switch. */
  const float s = c * m->inv_vs;
  if (m->p.index_mode == ORC_INDEX_TRUNC) return (int32_t)s;
  return (int32_t)floorf(s);
}

orc_map* orc_map_create(const orc_map_params* p) {
  orc_map* m = (orc_map*)calloc(1, sizeof(orc_map));
  m->p = *p;
  m->inv_vs = 1.0f / p->voxel_size;
  m->table_size = 1024;
  m->table = (int32_t*)malloc(m->table_size * sizeof(int32_t));
  for (size_t i = 0; i < m->table_size; i++) m->table[i] = -1;
  m->cap_vox = 512;
  m->vox = (voxel_t*)malloc(m->cap_vox * sizeof(voxel_t));
  for (int i = 0; i < 3; i++) { m->bb_min[i] = INFINITY; m->bb_max[i] = -INFINITY; }
  return m;
}

void orc_map_destroy(orc_map* m) {
  if (!m) return;
  for (size_t i = 0; i < m->n_vox; i++) { free(m->vox[i].xyz); free(m->vox[i].src); }
  free(m->vox);
  free(m->table);
  free(m);
}

static inline const voxel_t* map_find(const orc_map* m, int32_t kx, int32_t ky, int32_t kz) {
  const size_t mask = m->table_size - 1;
  size_t h = hash3(kx, ky, kz) & mask;
  for (;;) {
    const int32_t id = m->table[h];
    if (id < 0) return NULL;
    const voxel_t* v = &m->vox[id];
    if (v->k[0] == kx && v->k[1] == ky && v->k[2] == kz) return v;
    h = (h + 1) & mask;
  }
}

static void map_rehash(orc_map* m, size_t new_size) {
  free(m->table);
  m->table_size = new_size;
  m->table = (int32_t*)malloc(new_size * sizeof(int32_t));
  for (size_t i = 0; i < new_size; i++) m->table[i] = -1;
  const size_t mask = new_size - 1;
  for (size_t id = 0; id < m->n_vox; id++) {
    const voxel_t* v = &m->vox[id];
    size_t h = hash3(v->k[0], v->k[1], v->k[2]) & mask;
    while (m->table[h] >= 0) h = (h + 1) & mask;
    m->table[h] = (int32_t)id;
  }
}

static voxel_t* map_find_or_create(orc_map* m, int32_t kx, int32_t ky, int32_t kz) {
  size_t mask = m->table_size - 1;
  size_t h = hash3(kx, ky, kz) & mask;
  for (;;) {
    const int32_t id = m->table[h];
    if (id < 0) break;
    voxel_t* v = &m->vox[id];
    if (v->k[0] == kx && v->k[1] == ky && v->k[2] == kz) return v;
    h = (h + 1) & mask;
  }
  if ((m->n_vox + 1) * 2 > m->table_size) {
    map_rehash(m, m->table_size * 2);
    mask = m->table_size - 1;
    h = hash3(kx, ky, kz) & mask;
    while (m->table[h] >= 0) h = (h + 1) & mask;
  }
  if (m->n_vox == m->cap_vox) {
    m->cap_vox *= 2;
    m->vox = (voxel_t*)realloc(m->vox, m->cap_vox * sizeof(voxel_t));
  }
  voxel_t* v = &m->vox[m->n_vox];
  v->k[0] = kx; v->k[1] = ky; v->k[2] = kz;
  v->n = 0;
  v->ndt_plane = 0;
  v->ndt_dirty = 1;
  v->cap = m->p.max_points_per_voxel ? m->p.max_points_per_voxel : 8;
  v->xyz = (float*)malloc((size_t)v->cap * 3 * sizeof(float));
  v->src = (uint32_t*)malloc((size_t)v->cap * sizeof(uint32_t));
  m->table[h] = (int32_t)m->n_vox;
  m->n_vox++;
  return v;
}

void orc_map_insert(orc_map* m, const float* x, const float* y, const float* z, size_t n) {
  /* HashedVoxelPointCloud::insertPoint: drop if the voxel already holds max_points_per_voxel
   * (yaml:235); min_distance_between_points is 0 in both target pipelines (yaml:236). */
  for (size_t i = 0; i < n; i++) {
    const uint32_t src = (uint32_t)(m->n_offered + i);
    const float px = x[i], py = y[i], pz = z[i];
    if (!isfinite(px) || !isfinite(py) || !isfinite(pz)) continue;
    voxel_t* v = map_find_or_create(m, coord2idx(m, px), coord2idx(m, py), coord2idx(m, pz));
    if (m->p.max_points_per_voxel && v->n >= m->p.max_points_per_voxel) continue;
    if (m->p.min_distance_between_points > 0.f) {
      /* mola::NDT / HashedVoxelPointCloud insertOpts.min_distance_between_points (lidar3d-ndt.yaml:244) */
      const float md2 = m->p.min_distance_between_points * m->p.min_distance_between_points;
      int too_close = 0;
      for (uint32_t j = 0; j < v->n && !too_close; j++) {
        const float dx = v->xyz[3 * j] - px, dy = v->xyz[3 * j + 1] - py, dz = v->xyz[3 * j + 2] - pz;
        too_close = ((dx * dx + dy * dy) + dz * dz) < md2;
      }
      if (too_close) continue;
    }
    v->ndt_dirty = 1;
    if (v->n == v->cap) {
      v->cap *= 2;
      v->xyz = (float*)realloc(v->xyz, (size_t)v->cap * 3 * sizeof(float));
      v->src = (uint32_t*)realloc(v->src, (size_t)v->cap * sizeof(uint32_t));
    }
    v->xyz[3 * v->n + 0] = px; v->xyz[3 * v->n + 1] = py; v->xyz[3 * v->n + 2] = pz;
    v->src[v->n] = src;
    v->n++;
    m->n_points++;
    if (px < m->bb_min[0]) m->bb_min[0] = px;
    if (py < m->bb_min[1]) m->bb_min[1] = py;
    if (pz < m->bb_min[2]) m->bb_min[2] = pz;
    if (px > m->bb_max[0]) m->bb_max[0] = px;
    if (py > m->bb_max[1]) m->bb_max[1] = py;
    if (pz > m->bb_max[2]) m->bb_max[2] = pz;
  }
  m->n_offered += n;
}

size_t orc_map_num_points(const orc_map* m) { return m->n_points; }
size_t orc_map_num_voxels(const orc_map* m) {
  size_t c = 0; /* a voxel created by a dropped non-finite point cannot exist; all have n>=1 */
  for (size_t i = 0; i < m->n_vox; i++) c += (m->vox[i].n > 0);
  return c;
}
void orc_map_bbox(const orc_map* m, float mn[3], float mx[3]) {
  memcpy(mn, m->bb_min, sizeof(float) * 3);
  memcpy(mx, m->bb_max, sizeof(float) * 3);
}

static const orc_map* g_sort_map;
static int cmp_vox(const void* a, const void* b) {
  const voxel_t* va = &g_sort_map->vox[*(const uint32_t*)a];
  const voxel_t* vb = &g_sort_map->vox[*(const uint32_t*)b];
  for (int i = 0; i < 3; i++) {
    if (va->k[i] < vb->k[i]) return -1;
    if (va->k[i] > vb->k[i]) return 1;
  }
  return 0;
}

void orc_map_dump(const orc_map* m, float* x, float* y, float* z, uint32_t* src_idx, int32_t* vox_keys,
                  uint32_t* vox_first, uint32_t* vox_count) {
  uint32_t* order = (uint32_t*)malloc((m->n_vox + 1) * sizeof(uint32_t));
  size_t nv = 0;
  for (size_t i = 0; i < m->n_vox; i++)
    if (m->vox[i].n > 0) order[nv++] = (uint32_t)i;
  g_sort_map = m;
  qsort(order, nv, sizeof(uint32_t), cmp_vox);
  size_t o = 0;
  for (size_t i = 0; i < nv; i++) {
    const voxel_t* v = &m->vox[order[i]];
    if (vox_keys) { vox_keys[3 * i] = v->k[0]; vox_keys[3 * i + 1] = v->k[1]; vox_keys[3 * i + 2] = v->k[2]; }
    if (vox_first) vox_first[i] = (uint32_t)o;
    if (vox_count) vox_count[i] = v->n;
    for (uint32_t j = 0; j < v->n; j++, o++) {
      if (x) x[o] = v->xyz[3 * j];
      if (y) y[o] = v->xyz[3 * j + 1];
      if (z) z[o] = v->xyz[3 * j + 2];
      if (src_idx) src_idx[o] = v->src[j];
    }
  }
  free(order);
}

/* ---- NDT voxel statistics (mola::NDT role; SURVEY 8a row a13, App.B U10) ------------------ */
/* Cyclic Jacobi on a symmetric 3x3 (fp64), fixed sweep count; eigenvalues ascending in w[], eigenvectors in the
 * columns of V.  Written as a plain operation sequence so that the device code can follow it step by step. */
static void jacobi3(double a[3][3], double w[3], double V[3][3]) {
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) V[i][j] = (i == j) ? 1.0 : 0.0;
  for (int sweep = 0; sweep < 12; sweep++) {
    for (int pq = 0; pq < 3; pq++) {
      const int p = (pq == 2) ? 1 : 0, q = (pq == 0) ? 1 : 2, r = 3 - p - q;
      const double apq = a[p][q];
      if (apq == 0.0) continue;
      const double theta = (a[q][q] - a[p][p]) / (2.0 * apq);
      const double t = (theta >= 0.0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
      const double c = 1.0 / sqrt(t * t + 1.0), sn = t * c;
      const double app = a[p][p] - t * apq, aqq = a[q][q] + t * apq;
      const double arp = c * a[r][p] - sn * a[r][q], arq = sn * a[r][p] + c * a[r][q];
      a[p][p] = app; a[q][q] = aqq; a[p][q] = a[q][p] = 0.0;
      a[r][p] = a[p][r] = arp; a[r][q] = a[q][r] = arq;
      for (int i = 0; i < 3; i++) {
        const double vip = c * V[i][p] - sn * V[i][q], viq = sn * V[i][p] + c * V[i][q];
        V[i][p] = vip; V[i][q] = viq;
      }
    }
  }
  int idx[3] = {0, 1, 2};
  for (int i = 0; i < 3; i++) w[i] = a[i][i];
  for (int i = 0; i < 2; i++)
    for (int j = 0; j < 2 - i; j++)
      if (w[idx[j + 1]] < w[idx[j]]) { int t = idx[j]; idx[j] = idx[j + 1]; idx[j + 1] = t; }
  double ws[3], Vs[3][3];
  for (int k = 0; k < 3; k++) {
    ws[k] = w[idx[k]];
    for (int i = 0; i < 3; i++) Vs[i][k] = V[i][idx[k]];
  }
  for (int k = 0; k < 3; k++) {
    w[k] = ws[k];
    for (int i = 0; i < 3; i++) V[i][k] = Vs[i][k];
  }
}

static void voxel_update_ndt(const orc_map* m, voxel_t* v) {
  v->ndt_dirty = 0;
  v->ndt_plane = 0;
  v->ndt_c[0] = v->ndt_c[1] = v->ndt_c[2] = 0.f;
  v->ndt_n[0] = v->ndt_n[1] = v->ndt_n[2] = 0.f;
  const uint32_t minp = m->p.ndt_min_points ? m->p.ndt_min_points : 4;
  if (!(m->p.ndt_max_eigen_ratio > 0.f) || v->n < minp) return;
  double mu[3] = {0, 0, 0};
  for (uint32_t j = 0; j < v->n; j++)
    for (int a = 0; a < 3; a++) mu[a] += (double)v->xyz[3 * j + a];
  for (int a = 0; a < 3; a++) mu[a] /= (double)v->n;
  double C[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}};
  for (uint32_t j = 0; j < v->n; j++) {
    const double d[3] = {(double)v->xyz[3 * j] - mu[0], (double)v->xyz[3 * j + 1] - mu[1], (double)v->xyz[3 * j + 2] - mu[2]};
    for (int a = 0; a < 3; a++)
      for (int b = a; b < 3; b++) C[a][b] += d[a] * d[b];
  }
  for (int a = 0; a < 3; a++)
    for (int b = a; b < 3; b++) {
      C[a][b] /= (double)(v->n - 1);
      C[b][a] = C[a][b];
    }
  double w[3], V[3][3];
  jacobi3(C, w, V);
  for (int a = 0; a < 3; a++) v->ndt_c[a] = (float)mu[a];
  if (!(w[2] > 0.0) || !(w[0] / w[2] < (double)m->p.ndt_max_eigen_ratio)) return;
  double nrm[3] = {V[0][0], V[1][0], V[2][0]};
  const double len = sqrt(nrm[0] * nrm[0] + nrm[1] * nrm[1] + nrm[2] * nrm[2]);
  int big = 0;
  for (int a = 1; a < 3; a++)
    if (fabs(nrm[a]) > fabs(nrm[big])) big = a;
  const double sgn = (nrm[big] < 0.0 ? -1.0 : 1.0) / len;
  for (int a = 0; a < 3; a++) v->ndt_n[a] = (float)(nrm[a] * sgn);
  v->ndt_plane = 1;
}

static inline const voxel_t* voxel_ndt(const orc_map* m, const voxel_t* v) {
  if (v->ndt_dirty) {
#ifdef _OPENMP
#pragma omp critical(orc_ndt)
#endif
    if (v->ndt_dirty) voxel_update_ndt(m, (voxel_t*)v);
  }
  return v;
}

void orc_map_dump_ndt(const orc_map* m, float* cx, float* cy, float* cz, float* nx, float* ny, float* nz, uint32_t* is_plane) {
  uint32_t* order = (uint32_t*)malloc((m->n_vox + 1) * sizeof(uint32_t));
  size_t nv = 0;
  for (size_t i = 0; i < m->n_vox; i++)
    if (m->vox[i].n > 0) order[nv++] = (uint32_t)i;
  g_sort_map = m;
  qsort(order, nv, sizeof(uint32_t), cmp_vox);
  for (size_t i = 0; i < nv; i++) {
    const voxel_t* v = voxel_ndt(m, &m->vox[order[i]]);
    if (cx) cx[i] = v->ndt_c[0];
    if (cy) cy[i] = v->ndt_c[1];
    if (cz) cz[i] = v->ndt_c[2];
    if (nx) nx[i] = v->ndt_n[0];
    if (ny) ny[i] = v->ndt_n[1];
    if (nz) nz[i] = v->ndt_n[2];
    if (is_plane) is_plane[i] = v->ndt_plane;
  }
  free(order);
}

int orc_map_nn_single(const orc_map* m, float qx, float qy, float qz, float out_pt[3], float* out_d2,
                      uint32_t* out_src_idx, uint64_t* n_candidates, uint64_t* n_voxels_hit) {
  /* nn_single_search: 3x3x3 block around voxel(q), reach independent of the matcher threshold
   * (SURVEY App.B U3); scan order x outer, y middle, z inner (U2). */
  const int32_t cx = coord2idx(m, qx), cy = coord2idx(m, qy), cz = coord2idx(m, qz);
  float best = INFINITY;
  int found = 0;
  uint64_t nc = 0, nv = 0;
  for (int32_t ix = cx - 1; ix <= cx + 1; ix++)
    for (int32_t iy = cy - 1; iy <= cy + 1; iy++)
      for (int32_t iz = cz - 1; iz <= cz + 1; iz++) {
        const voxel_t* v = map_find(m, ix, iy, iz);
        if (!v || v->n == 0) continue;
        nv++;
        nc += v->n;
        for (uint32_t j = 0; j < v->n; j++) {
          const float dx = v->xyz[3 * j] - qx, dy = v->xyz[3 * j + 1] - qy, dz = v->xyz[3 * j + 2] - qz;
          const float d2 = (dx * dx + dy * dy) + dz * dz; /* fp32, un-fused, this order */
          if (d2 < best) {
            best = d2;
            found = 1;
            out_pt[0] = v->xyz[3 * j]; out_pt[1] = v->xyz[3 * j + 1]; out_pt[2] = v->xyz[3 * j + 2];
            *out_src_idx = v->src[j];
          }
        }
      }
  if (n_candidates) *n_candidates += nc;
  if (n_voxels_hit) *n_voxels_hit += nv;
  *out_d2 = best;
  return found;
}

/* ======================================================================================
 * Matcher -- Matcher_Points_Base::transform_local_to_global + Matcher_Points_DistanceThreshold
 * (yaml:195-204)
 * ==================================================================================== */
static inline void transform_pt(const double T[12], float lx, float ly, float lz, float* gx, float* gy, float* gz) {
  /* CPose3D::composePoint: double pose x float point, result rounded to float (U4) */
  const double x = lx, y = ly, z = lz;
  *gx = (float)(((R_(T, 0, 0) * x + R_(T, 0, 1) * y) + R_(T, 0, 2) * z) + t_(T, 0));
  *gy = (float)(((R_(T, 1, 0) * x + R_(T, 1, 1) * y) + R_(T, 1, 2) * z) + t_(T, 1));
  *gz = (float)(((R_(T, 2, 0) * x + R_(T, 2, 1) * y) + R_(T, 2, 2) * z) + t_(T, 2));
}

size_t orc_match_points(const orc_map* m, const float* lx, const float* ly, const float* lz, size_t n,
                        const double T[12], double threshold, double threshold_angular_deg,
                        uint32_t* local_idx, uint32_t* global_idx, float* gx, float* gy, float* gz, float* d2,
                        orc_match_stats* stats, int n_threads) {
  /* const float maxDistForCorrespondenceSquared = square(double threshold); same for the angle */
  const float thr2 = (float)(threshold * threshold);
  const double ang = threshold_angular_deg * 3.14159265358979323846 / 180.0;
  const float ang2 = (float)(ang * ang);
  uint8_t* ok = (uint8_t*)malloc(n ? n : 1);
  uint64_t nc = 0, nv = 0;
  (void)n_threads;
#ifdef _OPENMP
#pragma omp parallel for schedule(dynamic, 512) num_threads(n_threads > 0 ? n_threads : 1) reduction(+ : nc, nv)
#endif
  for (long i = 0; i < (long)n; i++) {
    float px, py, pz, q[3], dd;
    uint32_t gi = 0;
    transform_pt(T, lx[i], ly[i], lz[i], &px, &py, &pz);
    if (!isfinite(px) || !isfinite(py) || !isfinite(pz)) { ok[i] = 0; continue; } /* a non-finite point pairs with nothing */
    const int found = orc_map_nn_single(m, px, py, pz, q, &dd, &gi, &nc, &nv);
    const float norm2 = (px * px + py * py) + pz * pz;
    const float lim = thr2 + ang2 * norm2;
    ok[i] = (uint8_t)(found && dd < lim);
    if (ok[i]) { gx[i] = q[0]; gy[i] = q[1]; gz[i] = q[2]; d2[i] = dd; global_idx[i] = gi; }
  }
  /* ordered compaction (ascending local index), in place */
  size_t np = 0;
  for (size_t i = 0; i < n; i++)
    if (ok[i]) {
      local_idx[np] = (uint32_t)i; global_idx[np] = global_idx[i];
      gx[np] = gx[i]; gy[np] = gy[i]; gz[np] = gz[i]; d2[np] = d2[i];
      np++;
    }
  free(ok);
  if (stats) {
    stats->potential_pairings = n; /* pcLocal.size() * pairingsPerPoint, counted before any test (U6) */
    stats->n_candidates = nc;
    stats->n_voxels_hit = nv;
  }
  return np;
}

/* nn_multiple_search(q, k) [U] ("same scan keeping k best sorted"): the 3x3x3 block in the same order as
 * nn_single_search; a candidate enters the sorted list in front of the first entry it is STRICTLY nearer than, so among
 * equal distances the earlier scan position stays in front (k = 1 is nn_single_search).  Returns how many were found (<= k). */
int orc_map_nn_multiple(const orc_map* m, float qx, float qy, float qz, uint32_t k, float* out_pts /* 3 per entry */,
                        float* out_d2, uint32_t* out_src_idx) {
  const int32_t cx = coord2idx(m, qx), cy = coord2idx(m, qy), cz = coord2idx(m, qz);
  uint32_t found = 0;
  for (int32_t ix = cx - 1; ix <= cx + 1; ix++)
    for (int32_t iy = cy - 1; iy <= cy + 1; iy++)
      for (int32_t iz = cz - 1; iz <= cz + 1; iz++) {
        const voxel_t* v = map_find(m, ix, iy, iz);
        if (!v || v->n == 0) continue;
        for (uint32_t j = 0; j < v->n; j++) {
          const float dx = v->xyz[3 * j] - qx, dy = v->xyz[3 * j + 1] - qy, dz = v->xyz[3 * j + 2] - qz;
          const float d2 = (dx * dx + dy * dy) + dz * dz; /* fp32, un-fused, this order */
          uint32_t pos = found;
          while (pos > 0 && d2 < out_d2[pos - 1]) pos--;
          if (pos >= k) continue;
          const uint32_t last = found < k ? found : k - 1;
          for (uint32_t t = last; t > pos; t--) {
            out_d2[t] = out_d2[t - 1];
            out_src_idx[t] = out_src_idx[t - 1];
            out_pts[3 * t] = out_pts[3 * (t - 1)]; out_pts[3 * t + 1] = out_pts[3 * (t - 1) + 1]; out_pts[3 * t + 2] = out_pts[3 * (t - 1) + 2];
          }
          out_d2[pos] = d2;
          out_src_idx[pos] = v->src[j];
          out_pts[3 * pos] = v->xyz[3 * j]; out_pts[3 * pos + 1] = v->xyz[3 * j + 1]; out_pts[3 * pos + 2] = v->xyz[3 * j + 2];
          if (found < k) found++;
        }
      }
  return (int)found;
}

/* Matcher_Points_DistanceThreshold with pairingsPerPoint = k > 1 (rgbd.yaml:135-141): nn_multiple_search(k),
 * the neighbours taken in ascending distance while d^2 < thr^2 + ang^2 |p'|^2, "break at first failure".  Pairs in ascending
 * local index, a point's pairs in ascending distance.  Output arrays sized n * k. */
size_t orc_match_points_k(const orc_map* m, const float* lx, const float* ly, const float* lz, size_t n, const double T[12],
                          double threshold, double threshold_angular_deg, uint32_t k, uint32_t* local_idx,
                          uint32_t* global_idx, float* gx, float* gy, float* gz, float* d2, orc_match_stats* stats) {
  const float thr2 = (float)(threshold * threshold);
  const double ang = threshold_angular_deg * 3.14159265358979323846 / 180.0;
  const float ang2 = (float)(ang * ang);
  float* pts = (float*)malloc(sizeof(float) * 3 * (k ? k : 1));
  float* dd = (float*)malloc(sizeof(float) * (k ? k : 1));
  uint32_t* gi = (uint32_t*)malloc(sizeof(uint32_t) * (k ? k : 1));
  size_t np = 0;
  for (size_t i = 0; i < n; i++) {
    float px, py, pz;
    transform_pt(T, lx[i], ly[i], lz[i], &px, &py, &pz);
    if (!isfinite(px) || !isfinite(py) || !isfinite(pz)) continue; /* a non-finite point pairs with nothing */
    const int found = orc_map_nn_multiple(m, px, py, pz, k, pts, dd, gi);
    const float norm2 = (px * px + py * py) + pz * pz;
    const float lim = thr2 + ang2 * norm2;
    for (int r = 0; r < found; r++) {
      if (!(dd[r] < lim)) break;
      local_idx[np] = (uint32_t)i; global_idx[np] = gi[r];
      gx[np] = pts[3 * r]; gy[np] = pts[3 * r + 1]; gz[np] = pts[3 * r + 2]; d2[np] = dd[r];
      np++;
    }
  }
  free(pts); free(dd); free(gi);
  if (stats) {
    stats->potential_pairings = (uint64_t)n * k; /* pcLocal.size() * pairingsPerPoint, counted before any test (U6) */
    stats->n_candidates = 0;
    stats->n_voxels_hit = 0;
  }
  return np;
}

/* Matcher_Point2Plane on the NDT map (semantics: icp_oracle.h) */
size_t orc_match_pt2pl(const orc_map* m, const float* lx, const float* ly, const float* lz, size_t n, const double T[12],
                       double distance_threshold, uint32_t* local_idx, float* cx, float* cy, float* cz, float* nx,
                       float* ny, float* nz, int n_threads) {
  const float thr = (float)distance_threshold;
  uint8_t* ok = (uint8_t*)malloc(n ? n : 1);
  /* make sure every voxel's statistics are current before going parallel */
  for (size_t i = 0; i < m->n_vox; i++) voxel_ndt(m, &m->vox[i]);
  (void)n_threads;
#ifdef _OPENMP
#pragma omp parallel for schedule(dynamic, 512) num_threads(n_threads > 0 ? n_threads : 1)
#endif
  for (long i = 0; i < (long)n; i++) {
    float px, py, pz;
    transform_pt(T, lx[i], ly[i], lz[i], &px, &py, &pz);
    ok[i] = 0;
    if (!isfinite(px) || !isfinite(py) || !isfinite(pz)) continue;
    const int32_t vx = coord2idx(m, px), vy = coord2idx(m, py), vz = coord2idx(m, pz);
    float best = INFINITY;
    const voxel_t* bv = NULL;
    for (int32_t ix = vx - 1; ix <= vx + 1; ix++)
      for (int32_t iy = vy - 1; iy <= vy + 1; iy++)
        for (int32_t iz = vz - 1; iz <= vz + 1; iz++) {
          const voxel_t* v = map_find(m, ix, iy, iz);
          if (!v || !v->ndt_plane) continue;
          const float dx = v->ndt_c[0] - px, dy = v->ndt_c[1] - py, dz = v->ndt_c[2] - pz;
          const float d2 = (dx * dx + dy * dy) + dz * dz;
          if (d2 < best) { best = d2; bv = v; }
        }
    if (!bv) continue;
    const float dx = px - bv->ndt_c[0], dy = py - bv->ndt_c[1], dz = pz - bv->ndt_c[2];
    const float e = (bv->ndt_n[0] * dx + bv->ndt_n[1] * dy) + bv->ndt_n[2] * dz;
    /* SURVEY App.B U10: thr > 0 compares the point-to-plane distance; a NEGATIVE distance_threshold selects the other
     * candidate reading, distance to the plane's centroid: |p'-c|^2 < thr^2 (fp32, un-fused) */
    const int accept = thr < 0.f ? ((dx * dx + dy * dy) + dz * dz < thr * thr) : (fabsf(e) < thr);
    if (accept) {
      ok[i] = 1;
      cx[i] = bv->ndt_c[0]; cy[i] = bv->ndt_c[1]; cz[i] = bv->ndt_c[2];
      nx[i] = bv->ndt_n[0]; ny[i] = bv->ndt_n[1]; nz[i] = bv->ndt_n[2];
    }
  }
  size_t np = 0;
  for (size_t i = 0; i < n; i++)
    if (ok[i]) {
      local_idx[np] = (uint32_t)i;
      cx[np] = cx[i]; cy[np] = cy[i]; cz[np] = cz[i];
      nx[np] = nx[i]; ny[np] = ny[i]; nz[np] = nz[i];
      np++;
    }
  free(ok);
  return np;
}

/* Matcher_Point2Plane on a plain point map: k nearest neighbours + PCA (semantics: icp_oracle.h; rgbd.yaml:143-151) */
size_t orc_match_pt2pl_knn(const orc_map* m, const float* lx, const float* ly, const float* lz, size_t n, const double T[12],
                           const orc_pt2pl_knn_params* p, uint32_t* local_idx, float* cx, float* cy, float* cz, float* nx,
                           float* ny, float* nz) {
  const uint32_t k = p->knn ? p->knn : 1;
  const uint32_t min_pts = p->minimum_plane_points < 3 ? 3 : p->minimum_plane_points;
  const float r2 = (float)(p->search_radius * p->search_radius);
  float* pts = (float*)malloc(sizeof(float) * 3 * k);
  float* dd = (float*)malloc(sizeof(float) * k);
  uint32_t* gi = (uint32_t*)malloc(sizeof(uint32_t) * k);
  size_t np = 0;
  for (size_t i = 0; i < n; i++) {
    float px, py, pz;
    transform_pt(T, lx[i], ly[i], lz[i], &px, &py, &pz);
    if (!isfinite(px) || !isfinite(py) || !isfinite(pz)) continue;
    const int found = orc_map_nn_multiple(m, px, py, pz, k, pts, dd, gi);
    uint32_t cnt = 0;
    while ((int)cnt < found && dd[cnt] < r2) cnt++; /* ascending distances: the neighbours inside the radius are a prefix */
    if (cnt < min_pts) continue;
    double mu[3] = {0, 0, 0};
    for (uint32_t j = 0; j < cnt; j++)
      for (int a = 0; a < 3; a++) mu[a] += (double)pts[3 * j + a];
    for (int a = 0; a < 3; a++) mu[a] /= (double)cnt;
    double C[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}};
    for (uint32_t j = 0; j < cnt; j++) {
      const double d[3] = {(double)pts[3 * j] - mu[0], (double)pts[3 * j + 1] - mu[1], (double)pts[3 * j + 2] - mu[2]};
      for (int a = 0; a < 3; a++)
        for (int b = a; b < 3; b++) C[a][b] += d[a] * d[b];
    }
    for (int a = 0; a < 3; a++)
      for (int b = a; b < 3; b++) {
        C[a][b] /= (double)(cnt - 1);
        C[b][a] = C[a][b];
      }
    double w[3], V[3][3];
    jacobi3(C, w, V);
    if (!(w[2] > 0.0) || w[0] > p->plane_eigen_threshold * w[2]) continue;
    double nrm[3] = {V[0][0], V[1][0], V[2][0]};
    const double len = sqrt(nrm[0] * nrm[0] + nrm[1] * nrm[1] + nrm[2] * nrm[2]);
    int big = 0;
    for (int a = 1; a < 3; a++)
      if (fabs(nrm[a]) > fabs(nrm[big])) big = a;
    const double sgn = (nrm[big] < 0.0 ? -1.0 : 1.0) / len;
    for (int a = 0; a < 3; a++) nrm[a] *= sgn;
    const double dist = fabs((nrm[0] * ((double)px - mu[0]) + nrm[1] * ((double)py - mu[1])) + nrm[2] * ((double)pz - mu[2]));
    if (dist > p->distance_threshold) continue;
    local_idx[np] = (uint32_t)i;
    cx[np] = (float)mu[0]; cy[np] = (float)mu[1]; cz[np] = (float)mu[2];
    nx[np] = (float)nrm[0]; ny[np] = (float)nrm[1]; nz[np] = (float)nrm[2];
    np++;
  }
  free(pts); free(dd); free(gi);
  return np;
}

/* ======================================================================================
 * Solver -- optimal_tf_gauss_newton (Appendix A)
 * ==================================================================================== */
static inline double robust_weight(uint32_t kernel, double c, double e2) {
  switch (kernel) {
    case ORC_KERNEL_GM_C4: { const double c2 = c * c, d = c2 + e2; return (c2 * c2) / (d * d); }
    case ORC_KERNEL_GM_KISS: { const double d = c + e2; return (c * c) / (d * d); }
    case ORC_KERNEL_GM_BARRON: { const double d = e2 / (4.0 * c * c) + 1.0; return 1.0 / (d * d); }
    case ORC_KERNEL_CAUCHY: { const double c2 = c * c; return c2 / (c2 + e2); }
    case ORC_KERNEL_GM_C2: { const double c2 = c * c, d = c2 + e2; return c2 / (d * d); }
    default: return 1.0;
  }
}

typedef struct { double H[36]; double g[6]; double cost; } acc_t;

static inline void acc_row(acc_t* a, const double J[6], double e, double w) {
  for (int i = 0; i < 6; i++) {
    a->g[i] += w * J[i] * e;
    for (int j = 0; j < 6; j++) a->H[i * 6 + j] += w * J[i] * J[j];
  }
}

static void accumulate_range(const orc_pairs_pt2pt* pp, size_t p0, size_t p1, const orc_pairs_pt2pl* pl, size_t q0,
                             size_t q1, const orc_gn_params* p, const double T[12], acc_t* a) {
  memset(a, 0, sizeof(*a));
  for (size_t i = p0; i < p1; i++) {
    const double l[3] = {pp->lx[i], pp->ly[i], pp->lz[i]};
    const double q[3] = {pp->gx[i], pp->gy[i], pp->gz[i]};
    double e[3];
    for (int r = 0; r < 3; r++) e[r] = R_(T, r, 0) * l[0] + R_(T, r, 1) * l[1] + R_(T, r, 2) * l[2] + t_(T, r) - q[r];
    const double e2 = e[0] * e[0] + e[1] * e[1] + e[2] * e[2];
    const double w = p->weight_pt2pt * robust_weight(p->robust_kernel, p->robust_kernel_param, e2);
    /* J = [R | -R [l]x]  (3x6), right perturbation T*exp(eps) */
    for (int r = 0; r < 3; r++) {
      const double Rr[3] = {R_(T, r, 0), R_(T, r, 1), R_(T, r, 2)};
      double J[6];
      J[0] = Rr[0]; J[1] = Rr[1]; J[2] = Rr[2];
      /* -(Rr . [l]x): [l]x = [[0,-lz,ly],[lz,0,-lx],[-ly,lx,0]] */
      J[3] = -(Rr[1] * l[2] - Rr[2] * l[1]);
      J[4] = -(-Rr[0] * l[2] + Rr[2] * l[0]);
      J[5] = -(Rr[0] * l[1] - Rr[1] * l[0]);
      acc_row(a, J, e[r], w);
    }
    a->cost += w * e2;
  }
  if (pl)
    for (size_t i = q0; i < q1; i++) {
      const double l[3] = {pl->lx[i], pl->ly[i], pl->lz[i]};
      const double c[3] = {pl->cx[i], pl->cy[i], pl->cz[i]};
      const double nrm[3] = {pl->nx[i], pl->ny[i], pl->nz[i]};
      double g3[3];
      for (int r = 0; r < 3; r++) g3[r] = R_(T, r, 0) * l[0] + R_(T, r, 1) * l[1] + R_(T, r, 2) * l[2] + t_(T, r) - c[r];
      const double e = nrm[0] * g3[0] + nrm[1] * g3[1] + nrm[2] * g3[2];
      const double w = p->weight_pt2pl * robust_weight(p->robust_kernel, p->robust_kernel_param, e * e);
      /* J = n^T [R | -R[l]x] */
      double mloc[3]; /* R^T n */
      for (int j = 0; j < 3; j++) mloc[j] = R_(T, 0, j) * nrm[0] + R_(T, 1, j) * nrm[1] + R_(T, 2, j) * nrm[2];
      double J[6];
      J[0] = mloc[0]; J[1] = mloc[1]; J[2] = mloc[2];
      J[3] = l[1] * mloc[2] - l[2] * mloc[1];
      J[4] = l[2] * mloc[0] - l[0] * mloc[2];
      J[5] = l[0] * mloc[1] - l[1] * mloc[0];
      acc_row(a, J, e, w);
      a->cost += w * e * e;
    }
}

/* Eigen-style LDL^T with diagonal pivoting; x = -H^-1 g (pseudo-inverse on zero pivots).
 * Returns 0 if any pivot is non-finite. */
static int ldlt_solve6(const double Hin[36], const double b[6], double x[6]) {
  double A[36];
  memcpy(A, Hin, sizeof(A));
  int perm[6];
  for (int i = 0; i < 6; i++) perm[i] = i;
  for (int k = 0; k < 6; k++) {
    int piv = k;
    double best = fabs(A[k * 6 + k]);
    for (int i = k + 1; i < 6; i++)
      if (fabs(A[i * 6 + i]) > best) { best = fabs(A[i * 6 + i]); piv = i; }
    if (piv != k) {
      for (int j = 0; j < 6; j++) { double t = A[k * 6 + j]; A[k * 6 + j] = A[piv * 6 + j]; A[piv * 6 + j] = t; }
      for (int j = 0; j < 6; j++) { double t = A[j * 6 + k]; A[j * 6 + k] = A[j * 6 + piv]; A[j * 6 + piv] = t; }
      int t = perm[k]; perm[k] = perm[piv]; perm[piv] = t;
    }
    const double d = A[k * 6 + k];
    if (!isfinite(d)) return 0;
    if (fabs(d) > 2.2250738585072014e-308) {
      for (int i = k + 1; i < 6; i++) A[i * 6 + k] /= d;
      for (int i = k + 1; i < 6; i++)
        for (int j = k + 1; j <= i; j++) {
          A[i * 6 + j] -= A[i * 6 + k] * d * A[j * 6 + k];
          A[j * 6 + i] = A[i * 6 + j];
        }
    } else {
      for (int i = k + 1; i < 6; i++) A[i * 6 + k] = 0.0;
    }
  }
  double y[6];
  for (int i = 0; i < 6; i++) y[i] = b[perm[i]];
  for (int i = 0; i < 6; i++)
    for (int j = 0; j < i; j++) y[i] -= A[i * 6 + j] * y[j];
  for (int i = 0; i < 6; i++) {
    const double d = A[i * 6 + i];
    y[i] = (fabs(d) > 2.2250738585072014e-308) ? y[i] / d : 0.0;
  }
  for (int i = 5; i >= 0; i--)
    for (int j = i + 1; j < 6; j++) y[i] -= A[j * 6 + i] * y[j];
  for (int i = 0; i < 6; i++) x[perm[i]] = y[i];
  for (int i = 0; i < 6; i++)
    if (!isfinite(x[i])) return 0;
  return 1;
}

static void prior_term(const orc_prior* prior, const double T[12], double H[36], double g[6]) {
  /* e_p = log(T_prior^-1 (+) T) ; Jp = d e_p / d eps for T <- T*exp(eps), by central
   * differences (U9: exact derivative up to O(h^2)). */
  double Pinv[12], D[12], e0[6];
  orc_pose_inverse(prior->mean, Pinv);
  orc_pose_compose(Pinv, T, D);
  orc_se3_log(D, e0);
  double Jp[36];
  const double h = 1e-6;
  for (int j = 0; j < 6; j++) {
    double xi[6] = {0, 0, 0, 0, 0, 0}, E[12], Dp[12], ep[6], em[6];
    xi[j] = h;
    orc_se3_exp(xi, E); orc_pose_compose(D, E, Dp); orc_se3_log(Dp, ep);
    xi[j] = -h;
    orc_se3_exp(xi, E); orc_pose_compose(D, E, Dp); orc_se3_log(Dp, em);
    for (int i = 0; i < 6; i++) Jp[i * 6 + j] = (ep[i] - em[i]) / (2.0 * h);
  }
  double JtL[36]; /* Jp^T * Lambda */
  for (int i = 0; i < 6; i++)
    for (int j = 0; j < 6; j++) {
      double s = 0;
      for (int k = 0; k < 6; k++) s += Jp[k * 6 + i] * prior->info[k * 6 + j];
      JtL[i * 6 + j] = s;
    }
  for (int i = 0; i < 6; i++) {
    double s = 0;
    for (int k = 0; k < 6; k++) s += JtL[i * 6 + k] * e0[k];
    g[i] += s;
    for (int j = 0; j < 6; j++) {
      double h2 = 0;
      for (int k = 0; k < 6; k++) h2 += JtL[i * 6 + k] * Jp[k * 6 + j];
      H[i * 6 + j] += h2;
    }
  }
}

int orc_gn_solve(const orc_pairs_pt2pt* pp, const orc_pairs_pt2pl* pl, const orc_gn_params* p, const orc_prior* prior,
                 double T[12], orc_gn_step* trace, int n_threads) {
  const size_t np = pp ? pp->n : 0, nl = pl ? pl->n : 0;
  int nt = n_threads > 0 ? n_threads : 1;
  if ((size_t)nt > np + nl) nt = (int)((np + nl) ? (np + nl) : 1);
  acc_t* parts = (acc_t*)malloc(sizeof(acc_t) * (size_t)nt);
  int steps = 0;
  for (uint32_t it = 0; it < p->max_inner_iterations; it++) {
#ifdef _OPENMP
#pragma omp parallel for schedule(static, 1) num_threads(nt)
#endif
    for (int t = 0; t < nt; t++) {
      const size_t p0 = np * (size_t)t / (size_t)nt, p1 = np * (size_t)(t + 1) / (size_t)nt;
      const size_t q0 = nl * (size_t)t / (size_t)nt, q1 = nl * (size_t)(t + 1) / (size_t)nt;
      accumulate_range(pp, p0, p1, pl, q0, q1, p, T, &parts[t]);
    }
    acc_t a;
    memset(&a, 0, sizeof(a));
    for (int t = 0; t < nt; t++) { /* ordered join: deterministic for a given thread count */
      for (int i = 0; i < 36; i++) a.H[i] += parts[t].H[i];
      for (int i = 0; i < 6; i++) a.g[i] += parts[t].g[i];
      a.cost += parts[t].cost;
    }
    if (prior) prior_term(prior, T, a.H, a.g);
    if (trace) {
      memcpy(trace[it].H, a.H, sizeof(a.H));
      memcpy(trace[it].g, a.g, sizeof(a.g));
      trace[it].err_norm_sqr = a.cost;
      memset(trace[it].delta, 0, sizeof(trace[it].delta));
      memcpy(trace[it].T_after, T, sizeof(double) * 12);
    }
    if (sqrt(a.cost) <= p->max_cost) break; /* "target error" early exit (U8) */
    double x[6], delta[6];
    if (!ldlt_solve6(a.H, a.g, x)) { free(parts); return -1; }
    for (int i = 0; i < 6; i++) delta[i] = -x[i];
    double E[12];
    orc_se3_exp(delta, E);
    orc_pose_compose(T, E, T); /* T <- T (+) exp(delta) */
    steps++;
    if (trace) { memcpy(trace[it].delta, delta, sizeof(delta)); memcpy(trace[it].T_after, T, sizeof(double) * 12); }
    double dn = 0;
    for (int i = 0; i < 6; i++) dn += delta[i] * delta[i];
    if (sqrt(dn) < p->min_delta) break;
  }
  free(parts);
  return steps;
}

/* ======================================================================================
 * Covariance -- mp2p_icp::covariance
 * ==================================================================================== */
static int chol_inverse6(const double A[36], double Ainv[36]) {
  double L[36] = {0};
  for (int i = 0; i < 6; i++)
    for (int j = 0; j <= i; j++) {
      double s = A[i * 6 + j];
      for (int k = 0; k < j; k++) s -= L[i * 6 + k] * L[j * 6 + k];
      if (i == j) {
        if (!(s > 0.0)) return 0;
        L[i * 6 + i] = sqrt(s);
      } else
        L[i * 6 + j] = s / L[j * 6 + j];
    }
  for (int c = 0; c < 6; c++) {
    double y[6], x[6];
    for (int i = 0; i < 6; i++) {
      double s = (i == c) ? 1.0 : 0.0;
      for (int k = 0; k < i; k++) s -= L[i * 6 + k] * y[k];
      y[i] = s / L[i * 6 + i];
    }
    for (int i = 5; i >= 0; i--) {
      double s = y[i];
      for (int k = i + 1; k < 6; k++) s -= L[k * 6 + i] * x[k];
      x[i] = s / L[i * 6 + i];
    }
    for (int i = 0; i < 6; i++) Ainv[i * 6 + c] = x[i];
  }
  return 1;
}

void orc_covariance(const orc_pairs_pt2pt* pp, const orc_pairs_pt2pl* pl, const double T[12], double findif_xyz,
                    double findif_ang, double cov[36], double AtA_out[36]) {
  const size_t np = pp ? pp->n : 0, nl = pl ? pl->n : 0;
  memset(cov, 0, sizeof(double) * 36);
  if (np + nl == 0) {
    for (int i = 0; i < 6; i++) cov[i * 6 + i] = 1e6;
    if (AtA_out) memset(AtA_out, 0, sizeof(double) * 36);
    return;
  }
  double x0[6];
  orc_pose_to_ypr(T, x0);
  double Tp[6][12], Tm[6][12], hh[6];
  for (int j = 0; j < 6; j++) {
    hh[j] = (j < 3) ? findif_xyz : findif_ang;
    double x[6];
    memcpy(x, x0, sizeof(x)); x[j] += hh[j]; orc_pose_from_ypr(x, Tp[j]);
    memcpy(x, x0, sizeof(x)); x[j] -= hh[j]; orc_pose_from_ypr(x, Tm[j]);
  }
  double AtA[36] = {0};
  {
    /* point-to-point block rows, thread-split with an ordered join (same scheme as the GN accumulation) */
    int nt = orc_max_threads();
    if (nt > 32) nt = 32;
    if ((size_t)nt > np) nt = np ? (int)np : 1;
    double* parts = (double*)calloc((size_t)nt * 36, sizeof(double));
#ifdef _OPENMP
#pragma omp parallel for schedule(static, 1) num_threads(nt)
#endif
    for (int t = 0; t < nt; t++) {
      double* P = parts + (size_t)t * 36;
      const size_t i0 = np * (size_t)t / (size_t)nt, i1 = np * (size_t)(t + 1) / (size_t)nt;
      for (size_t i = i0; i < i1; i++) {
        const double l[3] = {pp->lx[i], pp->ly[i], pp->lz[i]};
        double A[3][6];
        for (int j = 0; j < 6; j++)
          for (int r = 0; r < 3; r++) {
            const double fp = R_(Tp[j], r, 0) * l[0] + R_(Tp[j], r, 1) * l[1] + R_(Tp[j], r, 2) * l[2] + t_(Tp[j], r);
            const double fm = R_(Tm[j], r, 0) * l[0] + R_(Tm[j], r, 1) * l[1] + R_(Tm[j], r, 2) * l[2] + t_(Tm[j], r);
            A[r][j] = (fp - fm) / (2.0 * hh[j]); /* the constant -q cancels */
          }
        for (int r = 0; r < 3; r++)
          for (int a = 0; a < 6; a++)
            for (int b = 0; b < 6; b++) P[a * 6 + b] += A[r][a] * A[r][b];
      }
    }
    for (int t = 0; t < nt; t++)
      for (int q = 0; q < 36; q++) AtA[q] += parts[(size_t)t * 36 + q];
    free(parts);
  }
  for (size_t i = 0; i < nl; i++) {
    const double l[3] = {pl->lx[i], pl->ly[i], pl->lz[i]};
    const double nrm[3] = {pl->nx[i], pl->ny[i], pl->nz[i]};
    double A[6];
    for (int j = 0; j < 6; j++) {
      double d = 0;
      for (int r = 0; r < 3; r++) {
        const double fp = R_(Tp[j], r, 0) * l[0] + R_(Tp[j], r, 1) * l[1] + R_(Tp[j], r, 2) * l[2] + t_(Tp[j], r);
        const double fm = R_(Tm[j], r, 0) * l[0] + R_(Tm[j], r, 1) * l[1] + R_(Tm[j], r, 2) * l[2] + t_(Tm[j], r);
        d += nrm[r] * (fp - fm);
      }
      A[j] = d / (2.0 * hh[j]);
    }
    for (int a = 0; a < 6; a++)
      for (int b = 0; b < 6; b++) AtA[a * 6 + b] += A[a] * A[b];
  }
  if (AtA_out) memcpy(AtA_out, AtA, sizeof(AtA));
  if (!chol_inverse6(AtA, cov)) {
    /* singular Hessian: no information in some direction -> report "unknown" like the
     * empty-pairings case */
    memset(cov, 0, sizeof(double) * 36);
    for (int i = 0; i < 6; i++) cov[i * 6 + i] = 1e6;
  }
}

/* ======================================================================================
 * ICP::align -- SURVEY 3.3 / 8a row a5 (call site LidarOdometry.cpp:961-962)
 * ==================================================================================== */
int orc_icp_align(const orc_map* m, const float* lx, const float* ly, const float* lz, size_t n,
                  const double T_guess[12], const orc_icp_params* p, const orc_prior* prior, orc_icp_result* res,
                  orc_icp_iter* trace, orc_pairs_out* final_pairs, int n_threads) {
  memset(res, 0, sizeof(*res));
  double T[12], Tprev[12];
  memcpy(T, T_guess, sizeof(T));
  memcpy(Tprev, T, sizeof(T));

  const size_t na = n ? n : 1;
  uint32_t* li = (uint32_t*)malloc(na * sizeof(uint32_t));
  uint32_t* gi = (uint32_t*)malloc(na * sizeof(uint32_t));
  float* buf = (float*)malloc(na * sizeof(float) * 16);
  float *gx = buf, *gy = buf + na, *gz = buf + 2 * na, *d2 = buf + 3 * na, *plx = buf + 4 * na, *ply = buf + 5 * na,
        *plz = buf + 6 * na;
  /* point-to-plane pairings (Matcher_Point2Plane, only when p->pt2pl_threshold is given) */
  float *qcx = buf + 7 * na, *qcy = buf + 8 * na, *qcz = buf + 9 * na, *qnx = buf + 10 * na, *qny = buf + 11 * na,
        *qnz = buf + 12 * na, *qlx = buf + 13 * na, *qly = buf + 14 * na, *qlz = buf + 15 * na;
  uint32_t* qli = (uint32_t*)malloc(na * sizeof(uint32_t));
  size_t npairs = 0, nplanes = 0;
  uint64_t potential = 0;
  res->termination_reason = ORC_TERM_UNDEFINED;

  uint32_t it;
  for (it = 0; it < p->max_iterations; it++) {
    /* ICP_ITERATION = it -> threshold / kernel param formulas (yaml:190,198) pre-evaluated */
    potential = 0;
    nplanes = 0;
    if (p->pt2pl_threshold) { /* matchers run in YAML order: Point2Plane first (lidar3d-ndt.yaml:195-210) */
      nplanes = orc_match_pt2pl(m, lx, ly, lz, n, T, p->pt2pl_threshold[it], qli, qcx, qcy, qcz, qnx, qny, qnz, n_threads);
      potential += n;
      for (size_t k = 0; k < nplanes; k++) { qlx[k] = lx[qli[k]]; qly[k] = ly[qli[k]]; qlz[k] = lz[qli[k]]; }
    }
    orc_match_stats st;
    npairs = orc_match_points(m, lx, ly, lz, n, T, p->threshold[it], p->threshold_angular_deg, li, gi, gx, gy, gz, d2,
                              &st, n_threads);
    potential += st.potential_pairings;
    res->n_candidates_total += st.n_candidates;
    if (p->pt2pt_skip_plane_paired && nplanes) {
      /* both index lists ascend: drop the point pairings of locals that already carry a plane pairing [U] (U12) */
      size_t w = 0, q = 0;
      for (size_t k = 0; k < npairs; k++) {
        while (q < nplanes && qli[q] < li[k]) q++;
        if (q < nplanes && qli[q] == li[k]) continue;
        li[w] = li[k]; gi[w] = gi[k]; gx[w] = gx[k]; gy[w] = gy[k]; gz[w] = gz[k]; d2[w] = d2[k];
        w++;
      }
      npairs = w;
    }
    if (npairs + nplanes == 0) { res->termination_reason = ORC_TERM_NO_PAIRINGS; break; }
    for (size_t k = 0; k < npairs; k++) { plx[k] = lx[li[k]]; ply[k] = ly[li[k]]; plz[k] = lz[li[k]]; }
    orc_pairs_pt2pt pp = {plx, ply, plz, gx, gy, gz, npairs};
    orc_pairs_pt2pl pl = {qlx, qly, qlz, qcx, qcy, qcz, qnx, qny, qnz, nplanes};
    orc_gn_params gp = p->gn;
    gp.robust_kernel_param = p->kernel_param[it];
    const int ok = orc_gn_solve(&pp, nplanes ? &pl : NULL, &gp, prior, T, NULL, n_threads);
    if (ok < 0) { res->termination_reason = ORC_TERM_SOLVER_ERROR; break; }
    /* stall test on log(T_prev^-1 (+) T_new) (yaml:174-175) */
    double Pinv[12], D[12], d[6];
    orc_pose_inverse(Tprev, Pinv);
    orc_pose_compose(Pinv, T, D);
    orc_se3_log(D, d);
    const double dtr = sqrt(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
    const double drot = sqrt(d[3] * d[3] + d[4] * d[4] + d[5] * d[5]);
    if (trace) {
      memcpy(trace[it].T, T, sizeof(T));
      trace[it].n_pairs = (uint32_t)(npairs + nplanes);
      trace[it].threshold = p->threshold[it];
      trace[it].kernel_param = p->kernel_param[it];
      trace[it].delta_trans = dtr;
      trace[it].delta_rot = drot;
    }
    if (!p->disable_stall_test && dtr < p->min_abs_step_trans && drot < p->min_abs_step_rot) {
      res->termination_reason = ORC_TERM_STALLED;
      break;
    }
    if (p->hook_enabled) {
      /* LidarOdometry.cpp:932-949 */
      double Cinv[12], S[12], w[3];
      orc_pose_inverse(p->hook_checkpoint, Cinv);
      orc_pose_compose(Cinv, T, S);
      orc_so3_log(S, w);
      const double ht = sqrt(t_(S, 0) * t_(S, 0) + t_(S, 1) * t_(S, 1) + t_(S, 2) * t_(S, 2));
      const double hr = sqrt(w[0] * w[0] + w[1] * w[1] + w[2] * w[2]);
      if (ht > p->hook_min_trans || hr > p->hook_min_rot) { res->termination_reason = ORC_TERM_HOOK_REQUEST; break; }
    }
    memcpy(Tprev, T, sizeof(T));
  }
  res->n_iterations = it;
  if (it >= p->max_iterations) res->termination_reason = ORC_TERM_MAX_ITERATIONS;

  memcpy(res->T, T, sizeof(T));
  res->n_final_pairs = (uint32_t)(npairs + nplanes);
  res->n_final_pairs_pt2pl = (uint32_t)nplanes;
  res->potential_pairings = potential;
  res->quality = ((npairs + nplanes) && potential) ? (double)(npairs + nplanes) / (double)potential : 0.0; /* PairedRatio */
  if (p->compute_covariance) {
    orc_pairs_pt2pt pp = {plx, ply, plz, gx, gy, gz, npairs};
    orc_pairs_pt2pl pl = {qlx, qly, qlz, qcx, qcy, qcz, qnx, qny, qnz, nplanes};
    orc_covariance(&pp, nplanes ? &pl : NULL, T, p->cov_findif_xyz, p->cov_findif_ang, res->cov, NULL);
  }
  if (final_pairs) {
    for (size_t k = 0; k < npairs; k++) {
      if (final_pairs->local_idx) final_pairs->local_idx[k] = li[k];
      if (final_pairs->global_idx) final_pairs->global_idx[k] = gi[k];
      if (final_pairs->gx) final_pairs->gx[k] = gx[k];
      if (final_pairs->gy) final_pairs->gy[k] = gy[k];
      if (final_pairs->gz) final_pairs->gz[k] = gz[k];
      if (final_pairs->d2) final_pairs->d2[k] = d2[k];
    }
  }
  free(li); free(gi); free(buf); free(qli);
  return 0;
}

/* ======================================================================================
 * SURVEY 8(f) rows f1 / f2
 * ==================================================================================== */
static void map_recompute_bbox(orc_map* m) {
  for (int i = 0; i < 3; i++) { m->bb_min[i] = INFINITY; m->bb_max[i] = -INFINITY; }
  for (size_t id = 0; id < m->n_vox; id++) {
    const voxel_t* v = &m->vox[id];
    for (uint32_t j = 0; j < v->n; j++)
      for (int a = 0; a < 3; a++) {
        const float c = v->xyz[3 * j + a];
        if (c < m->bb_min[a]) m->bb_min[a] = c;
        if (c > m->bb_max[a]) m->bb_max[a] = c;
      }
  }
}

void orc_map_insert_posed(orc_map* m, const float* x, const float* y, const float* z, size_t n, const double T[12],
                          float remove_voxels_farther_than) {
  float* g = (float*)malloc((n ? n : 1) * 3 * sizeof(float));
  for (size_t i = 0; i < n; i++) transform_pt(T, x[i], y[i], z[i], &g[i], &g[n + i], &g[2 * n + i]);
  orc_map_insert(m, g, g + n, g + 2 * n, n);
  free(g);
  if (remove_voxels_farther_than > 0.f) {
    const int dist_in_grid = (int)ceilf(remove_voxels_farther_than * m->inv_vs);
    const int32_t c[3] = {coord2idx(m, (float)t_(T, 0)), coord2idx(m, (float)t_(T, 1)), coord2idx(m, (float)t_(T, 2))};
    int removed = 0;
    for (size_t id = 0; id < m->n_vox; id++) {
      voxel_t* v = &m->vox[id];
      if (!v->n) continue;
      const long long d0 = llabs((long long)v->k[0] - c[0]), d1 = llabs((long long)v->k[1] - c[1]),
                      d2 = llabs((long long)v->k[2] - c[2]);
      int far;
      if (m->p.far_voxel_metric == 1) far = d0 + d1 + d2 > dist_in_grid;                                     /* L1 */
      else if (m->p.far_voxel_metric == 2) far = d0 * d0 + d1 * d1 + d2 * d2 > (long long)dist_in_grid * dist_in_grid; /* L2 */
      else far = (d0 > d1 ? (d0 > d2 ? d0 : d2) : (d1 > d2 ? d1 : d2)) > dist_in_grid;                       /* Chebyshev */
      if (far) { /* erase(): an emptied voxel behaves like a missing one everywhere */
        m->n_points -= v->n;
        v->n = 0;
        v->ndt_dirty = 1;
        removed = 1;
      }
    }
    if (removed) map_recompute_bbox(m);
  }
}

void orc_adjust_timestamps(float* t, size_t n, int method, float time_offset) {
  if (!n || method == ORC_TS_NONE) return;
  float tmin = t[0], tmax = t[0];
  for (size_t i = 1; i < n; i++) {
    if (t[i] < tmin) tmin = t[i];
    if (t[i] > tmax) tmax = t[i];
  }
  const float dt = (method == ORC_TS_MIDDLE_IS_ZERO) ? 0.5f * (tmin + tmax) : tmin;
  for (size_t i = 0; i < n; i++) t[i] = (t[i] - dt) + time_offset;
}

typedef struct { int32_t k[3]; uint32_t used; } dec_cell;

size_t orc_decimate_first_point(const float* x, const float* y, const float* z, size_t n, float resolution,
                                uint32_t min_points_to_filter, int index_mode, uint32_t* out_idx) {
  size_t o = 0;
  if (resolution <= 0.f || n < min_points_to_filter) {
    for (size_t i = 0; i < n; i++)
      if (isfinite(x[i]) && isfinite(y[i]) && isfinite(z[i])) out_idx[o++] = (uint32_t)i;
    return o;
  }
  const float inv = 1.0f / resolution;
  size_t tsize = 64;
  while (tsize < 2 * n) tsize <<= 1;
  dec_cell* tab = (dec_cell*)calloc(tsize, sizeof(dec_cell));
  for (size_t i = 0; i < n; i++) {
    if (!isfinite(x[i]) || !isfinite(y[i]) || !isfinite(z[i])) continue;
    int32_t k[3];
    const float s[3] = {x[i] * inv, y[i] * inv, z[i] * inv};
    for (int a = 0; a < 3; a++) k[a] = (index_mode == ORC_INDEX_TRUNC) ? (int32_t)s[a] : (int32_t)floorf(s[a]);
    size_t h = hash3(k[0], k[1], k[2]) & (tsize - 1);
    int seen = 0;
    while (tab[h].used) {
      if (tab[h].k[0] == k[0] && tab[h].k[1] == k[1] && tab[h].k[2] == k[2]) { seen = 1; break; }
      h = (h + 1) & (tsize - 1);
    }
    if (seen) continue; /* the voxel already holds its first point */
    tab[h].used = 1;
    tab[h].k[0] = k[0]; tab[h].k[1] = k[1]; tab[h].k[2] = k[2];
    out_idx[o++] = (uint32_t)i;
  }
  free(tab);
  return o;
}

typedef struct { int32_t k[3]; uint32_t used, count, best; float sx, sy, sz, mx, my, mz, best_e; } cta_cell;

size_t orc_decimate_closest_to_average(const float* x, const float* y, const float* z, size_t n, float resolution,
                                       uint32_t min_points_to_filter, int index_mode, uint32_t* out_idx) {
  size_t o = 0;
  if (resolution <= 0.f || n < min_points_to_filter) {
    for (size_t i = 0; i < n; i++)
      if (isfinite(x[i]) && isfinite(y[i]) && isfinite(z[i])) out_idx[o++] = (uint32_t)i;
    return o;
  }
  const float inv = 1.0f / resolution;
  size_t tsize = 64;
  while (tsize < 2 * n) tsize <<= 1;
  cta_cell* tab = (cta_cell*)calloc(tsize, sizeof(cta_cell));
  uint32_t* cell_of = (uint32_t*)malloc((n ? n : 1) * sizeof(uint32_t));
  /* pass 1: the voxel of every point; per voxel the float sum of its points in input order */
  for (size_t i = 0; i < n; i++) {
    cell_of[i] = 0xFFFFFFFFu;
    if (!isfinite(x[i]) || !isfinite(y[i]) || !isfinite(z[i])) continue;
    int32_t k[3];
    const float s[3] = {x[i] * inv, y[i] * inv, z[i] * inv};
    for (int a = 0; a < 3; a++) k[a] = (index_mode == ORC_INDEX_TRUNC) ? (int32_t)s[a] : (int32_t)floorf(s[a]);
    size_t h = hash3(k[0], k[1], k[2]) & (tsize - 1);
    while (tab[h].used && !(tab[h].k[0] == k[0] && tab[h].k[1] == k[1] && tab[h].k[2] == k[2])) h = (h + 1) & (tsize - 1);
    if (!tab[h].used) {
      tab[h].used = 1;
      tab[h].k[0] = k[0]; tab[h].k[1] = k[1]; tab[h].k[2] = k[2];
    }
    tab[h].sx += x[i];
    tab[h].sy += y[i];
    tab[h].sz += z[i];
    tab[h].count++;
    cell_of[i] = (uint32_t)h;
  }
  for (size_t h = 0; h < tsize; h++)
    if (tab[h].used) {
      const float inv_n = 1.0f / (float)tab[h].count;
      tab[h].mx = tab[h].sx * inv_n;
      tab[h].my = tab[h].sy * inv_n;
      tab[h].mz = tab[h].sz * inv_n;
      tab[h].count = 0; /* (now: points seen by pass 2) */
    }
  /* pass 2: per voxel the point closest to the mean; a strictly smaller error replaces the candidate (the first of equals stays) */
  for (size_t i = 0; i < n; i++) {
    if (cell_of[i] == 0xFFFFFFFFu) continue;
    cta_cell* c = &tab[cell_of[i]];
    const float dx = x[i] - c->mx, dy = y[i] - c->my, dz = z[i] - c->mz;
    const float e = (dx * dx + dy * dy) + dz * dz;
    if (c->count == 0 || e < c->best_e) {
      c->best_e = e;
      c->best = (uint32_t)i;
    }
    c->count++;
  }
  for (size_t i = 0; i < n; i++)
    if (cell_of[i] != 0xFFFFFFFFu && tab[cell_of[i]].best == (uint32_t)i) out_idx[o++] = (uint32_t)i;
  free(cell_of);
  free(tab);
  return o;
}

size_t orc_filter_by_range(const float* x, const float* y, const float* z, size_t n, float range_min, float range_max,
                           const float center[3], uint32_t* out_idx) {
  const float sq_min = range_min * range_min, sq_max = range_max * range_max;
  size_t o = 0;
  for (size_t i = 0; i < n; i++) {
    const float dx = x[i] - center[0], dy = y[i] - center[1], dz = z[i] - center[2];
    const float sq = (dx * dx + dy * dy) + dz * dz;
    if (sq >= sq_min && sq <= sq_max) out_idx[o++] = (uint32_t)i;
  }
  return o;
}

size_t orc_filter_bbox(const float* x, const float* y, const float* z, size_t n, const float bb_min[3],
                       const float bb_max[3], int keep_inside, uint32_t* out_idx) {
  size_t o = 0;
  for (size_t i = 0; i < n; i++) {
    const int inside = x[i] >= bb_min[0] && x[i] <= bb_max[0] && y[i] >= bb_min[1] && y[i] <= bb_max[1] &&
                       z[i] >= bb_min[2] && z[i] <= bb_max[2];
    if ((inside != 0) == (keep_inside != 0)) out_idx[o++] = (uint32_t)i;
  }
  return o;
}

void orc_deskew(const float* x, const float* y, const float* z, const float* t, size_t n, const double twist[6],
                float* ox, float* oy, float* oz) {
  for (size_t i = 0; i < n; i++) {
    const double dt = (double)t[i];
    const double xi[6] = {0.0, 0.0, 0.0, twist[3] * dt, twist[4] * dt, twist[5] * dt};
    double T[12];
    orc_se3_exp(xi, T); /* zero translation part: pure Exp_SO3(w dt) */
    T[3] = twist[0] * dt; T[7] = twist[1] * dt; T[11] = twist[2] * dt;
    transform_pt(T, x[i], y[i], z[i], &ox[i], &oy[i], &oz[i]);
  }
}

void orc_preprocess(const float* x, const float* y, const float* z, size_t n, const orc_preprocess_params* p,
                    uint32_t* idx_map, size_t* n_map, uint32_t* idx_icp, size_t* n_icp) {
  /* every stage materialises its layer like the filter classes do, carrying the raw index along */
  float* bx = (float*)malloc((n ? n : 1) * 3 * sizeof(float));
  float *by = bx + n, *bz = bx + 2 * n;
  uint32_t* cur = (uint32_t*)malloc((n ? n : 1) * sizeof(uint32_t));
  uint32_t* sel = (uint32_t*)malloc((n ? n : 1) * sizeof(uint32_t));
  size_t m = n;
  for (size_t i = 0; i < n; i++) { cur[i] = (uint32_t)i; bx[i] = x[i]; by[i] = y[i]; bz[i] = z[i]; }
#define ORC_APPLY(count_expr)                                                        \
  do {                                                                               \
    const size_t k_ = (count_expr);                                                  \
    for (size_t i = 0; i < k_; i++) {                                                \
      const uint32_t s_ = sel[i];                                                    \
      cur[i] = cur[s_]; bx[i] = bx[s_]; by[i] = by[s_]; bz[i] = bz[s_];              \
    }                                                                                \
    m = k_;                                                                          \
  } while (0)
  /* (selection lists are ascending, so the in-place gather never overwrites an unread entry) */
  ORC_APPLY(p->decim_map_method == ORC_DECIMATE_CLOSEST_TO_AVERAGE
                ? orc_decimate_closest_to_average(bx, by, bz, m, p->decim_map_resolution, p->min_points_to_filter, p->index_mode, sel)
                : orc_decimate_first_point(bx, by, bz, m, p->decim_map_resolution, p->min_points_to_filter, p->index_mode, sel));
  if (p->range_max > 0.f) ORC_APPLY(orc_filter_by_range(bx, by, bz, m, p->range_min, p->range_max, p->range_center, sel));
  if (p->bbox_mode) ORC_APPLY(orc_filter_bbox(bx, by, bz, m, p->bbox_min, p->bbox_max, p->bbox_mode == 2, sel));
  for (size_t i = 0; i < m; i++) idx_map[i] = cur[i];
  *n_map = m;
  ORC_APPLY(p->decim_icp_method == ORC_DECIMATE_CLOSEST_TO_AVERAGE
                ? orc_decimate_closest_to_average(bx, by, bz, m, p->decim_icp_resolution, p->min_points_to_filter, p->index_mode, sel)
                : orc_decimate_first_point(bx, by, bz, m, p->decim_icp_resolution, p->min_points_to_filter, p->index_mode, sel));
  for (size_t i = 0; i < m; i++) idx_icp[i] = cur[i];
  *n_icp = m;
#undef ORC_APPLY
  free(bx); free(cur); free(sel);
}

int orc_max_threads(void) {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}
