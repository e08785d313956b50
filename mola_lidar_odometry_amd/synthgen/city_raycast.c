// city_raycast.c -- DATA GENERATION ONLY (neither product path nor oracle): ray-casts the sweeps of a spinning LiDAR
// through a synthetic city of axis-aligned boxes (buildings, cars) and capped vertical cylinders (poles, trunks, tree
// crowns) standing on the ground plane z = 0.  There is no dataset in either container, so the long
// odometry drive behind bench.py's `single_sequence*` extras and tests/test_gpu_accuracy.py is generated with this.
//
// The sensor moves with a constant twist while it scans (the motion FilterDeskew undoes, lidar3d-default.yaml:328-350
// of the reference): azimuth column j fires at t_j = (j / azimuths - 0.5) * sweep_time from the pose
// T_ref (+) (Exp_SO3(w t_j), v t_j); points are reported in the sensor frame AT FIRING TIME (skewed), ring-major.
//
// Acceleration: a uniform 2-D grid over x/y with the primitives overlapping every cell, walked by a DDA.
// Deterministic: the range noise of ray i is a counter-based hash of (seed, i), so the output does not depend on the
// thread count.
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef struct {
  int nb, nc;
  float* boxes;  // nb x 6: xmin ymin zmin xmax ymax zmax
  float* cyls;   // nc x 5: cx cy r zmin zmax
  double x0, y0, cell;
  int gx, gy;
  int* start;  // gx*gy + 1
  int* items;  // >= 0: box index; < 0: cylinder ~index
} City;

static void prim_bounds(const City* c, int item, double* b) {
  if (item >= 0) {
    const float* q = c->boxes + 6 * (size_t)item;
    b[0] = q[0], b[1] = q[1], b[2] = q[3], b[3] = q[4];
  } else {
    const float* q = c->cyls + 5 * (size_t)(~item);
    b[0] = q[0] - q[2], b[1] = q[1] - q[2], b[2] = q[0] + q[2], b[3] = q[1] + q[2];
  }
}

static int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

__attribute__((visibility("default"))) void* city_create(const float* boxes, int nb, const float* cyls, int nc, double xmin,
                                                          double ymin, double xmax, double ymax, double cell) {
  City* c = (City*)calloc(1, sizeof(City));
  c->nb = nb, c->nc = nc;
  c->boxes = (float*)malloc(sizeof(float) * 6 * (size_t)(nb > 0 ? nb : 1));
  c->cyls = (float*)malloc(sizeof(float) * 5 * (size_t)(nc > 0 ? nc : 1));
  if (nb) memcpy(c->boxes, boxes, sizeof(float) * 6 * (size_t)nb);
  if (nc) memcpy(c->cyls, cyls, sizeof(float) * 5 * (size_t)nc);
  c->x0 = xmin, c->y0 = ymin, c->cell = cell;
  c->gx = (int)ceil((xmax - xmin) / cell), c->gy = (int)ceil((ymax - ymin) / cell);
  if (c->gx < 1) c->gx = 1;
  if (c->gy < 1) c->gy = 1;
  const size_t ncell = (size_t)c->gx * (size_t)c->gy;
  int* count = (int*)calloc(ncell + 1, sizeof(int));
  for (int pass = 0; pass < 2; pass++) {
    if (pass == 1) {
      c->start = (int*)malloc(sizeof(int) * (ncell + 1));
      int acc = 0;
      for (size_t k = 0; k < ncell; k++) {
        c->start[k] = acc;
        acc += count[k];
        count[k] = 0;
      }
      c->start[ncell] = acc;
      c->items = (int*)malloc(sizeof(int) * (size_t)(acc > 0 ? acc : 1));
    }
    for (int it = -nc; it < nb; it++) {
      double b[4];
      prim_bounds(c, it, b);
      const int ix0 = clampi((int)floor((b[0] - xmin) / cell), 0, c->gx - 1), ix1 = clampi((int)floor((b[2] - xmin) / cell), 0, c->gx - 1);
      const int iy0 = clampi((int)floor((b[1] - ymin) / cell), 0, c->gy - 1), iy1 = clampi((int)floor((b[3] - ymin) / cell), 0, c->gy - 1);
      for (int iy = iy0; iy <= iy1; iy++)
        for (int ix = ix0; ix <= ix1; ix++) {
          const size_t k = (size_t)iy * (size_t)c->gx + (size_t)ix;
          if (pass == 1) c->items[c->start[k] + count[k]] = it;
          count[k]++;
        }
    }
  }
  free(count);
  return c;
}

__attribute__((visibility("default"))) void city_destroy(void* h) {
  City* c = (City*)h;
  if (!c) return;
  free(c->boxes), free(c->cyls), free(c->start), free(c->items), free(c);
}

static double hit_box(const float* q, const double o[3], const double d[3]) {
  double tn = -INFINITY, tf = INFINITY;
  for (int a = 0; a < 3; a++) {
    if (d[a] == 0.0) {
      if (o[a] < q[a] || o[a] > q[a + 3]) return INFINITY;
      continue;
    }
    double t0 = (q[a] - o[a]) / d[a], t1 = (q[a + 3] - o[a]) / d[a];
    if (t0 > t1) {
      const double s = t0;
      t0 = t1, t1 = s;
    }
    if (t0 > tn) tn = t0;
    if (t1 < tf) tf = t1;
  }
  return (tf >= tn && tn > 1e-6) ? tn : INFINITY;
}

static double hit_cyl(const float* q, const double o[3], const double d[3]) {
  const double cx = q[0], cy = q[1], r = q[2], z0 = q[3], z1 = q[4];
  double best = INFINITY;
  const double ox = o[0] - cx, oy = o[1] - cy;
  const double a = d[0] * d[0] + d[1] * d[1];
  if (a > 1e-18) {
    const double b = ox * d[0] + oy * d[1], cc = ox * ox + oy * oy - r * r;
    const double disc = b * b - a * cc;
    if (disc >= 0.0) {
      const double t = (-b - sqrt(disc)) / a;
      if (t > 1e-6) {
        const double z = o[2] + t * d[2];
        if (z >= z0 && z <= z1) best = t;
      }
    }
  }
  if (d[2] != 0.0) {  // caps
    for (int k = 0; k < 2; k++) {
      const double zc = k ? z1 : z0;
      if (zc <= 0.0) continue;
      const double t = (zc - o[2]) / d[2];
      if (t > 1e-6 && t < best) {
        const double x = ox + t * d[0], y = oy + t * d[1];
        if (x * x + y * y <= r * r) best = t;
      }
    }
  }
  return best;
}

// first surface along the ray within tmax (INFINITY = none)
static double cast(const City* c, const double o[3], const double d[3], double tmax) {
  double best = INFINITY;
  if (d[2] < 0.0 && o[2] > 0.0) {
    const double tg = -o[2] / d[2];
    if (tg < tmax) best = tg, tmax = tg;
  }
  const double cell = c->cell;
  double fx = (o[0] - c->x0) / cell, fy = (o[1] - c->y0) / cell;
  int ix = (int)floor(fx), iy = (int)floor(fy);
  if (ix < 0 || iy < 0 || ix >= c->gx || iy >= c->gy) return best;  // (the route stays inside the grid)
  const int sx = d[0] > 0 ? 1 : -1, sy = d[1] > 0 ? 1 : -1;
  const double inv_x = d[0] != 0.0 ? 1.0 / d[0] : INFINITY, inv_y = d[1] != 0.0 ? 1.0 / d[1] : INFINITY;
  double tx = d[0] != 0.0 ? ((c->x0 + (ix + (sx > 0)) * cell) - o[0]) * inv_x : INFINITY;
  double ty = d[1] != 0.0 ? ((c->y0 + (iy + (sy > 0)) * cell) - o[1]) * inv_y : INFINITY;
  const double dtx = fabs(cell * inv_x), dty = fabs(cell * inv_y);
  double t_enter = 0.0;
  while (t_enter <= tmax && t_enter <= best) {
    const size_t k = (size_t)iy * (size_t)c->gx + (size_t)ix;
    for (int p = c->start[k]; p < c->start[k + 1]; p++) {
      const int it = c->items[p];
      const double t = it >= 0 ? hit_box(c->boxes + 6 * (size_t)it, o, d) : hit_cyl(c->cyls + 5 * (size_t)(~it), o, d);
      if (t < best) best = t;
    }
    if (tx < ty) {
      t_enter = tx, tx += dtx, ix += sx;
    } else {
      t_enter = ty, ty += dty, iy += sy;
    }
    if (ix < 0 || iy < 0 || ix >= c->gx || iy >= c->gy) break;
  }
  return best <= tmax ? best : INFINITY;
}

static int g_threads = 0;  // 0 = OpenMP's default
// The ray loop of one sweep is ~10 ms of work: more than a few dozen threads only add fork/join cost (a 256-thread box
// cast a sweep 8x slower with all of them than with 16).
__attribute__((visibility("default"))) void city_set_threads(int n) { g_threads = n > 0 ? n : 0; }

static uint64_t splitmix(uint64_t x) {
  x += 0x9E3779B97F4A7C15ull;
  x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
  x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
  return x ^ (x >> 31);
}

static double gauss(uint64_t seed, uint64_t i) {
  const uint64_t a = splitmix(seed * 0x100000001B3ull + 2 * i), b = splitmix(seed * 0x100000001B3ull + 2 * i + 1);
  const double u1 = ((double)(a >> 11) + 1.0) / 9007199254740993.0, u2 = (double)(b >> 11) / 9007199254740992.0;
  return sqrt(-2.0 * log(u1)) * cos(6.283185307179586 * u2);
}

static void so3_exp(const double w[3], double R[9]) {
  const double th2 = w[0] * w[0] + w[1] * w[1] + w[2] * w[2], th = sqrt(th2);
  const double a = th < 1e-8 ? 1.0 - th2 / 6.0 : sin(th) / th, b = th < 1e-8 ? 0.5 - th2 / 24.0 : (1.0 - cos(th)) / th2;
  const double W[9] = {0, -w[2], w[1], w[2], 0, -w[0], -w[1], w[0], 0};
  double W2[9];
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) W2[3 * i + j] = W[3 * i] * W[j] + W[3 * i + 1] * W[3 + j] + W[3 * i + 2] * W[6 + j];
  for (int k = 0; k < 9; k++) R[k] = (k % 4 == 0 ? 1.0 : 0.0) + a * W[k] + b * W2[k];
}

// One sweep.  xyz_out: rings*azimuths*3 floats, t_out: rings*azimuths floats (worst case).  Returns the number of returns.
__attribute__((visibility("default"))) int city_sweep(const void* h, const double T[12], const double twist[6], double sweep_time, int rings,
                                                       int azimuths, double el_top_deg, double el_bottom_deg, uint64_t seed, double max_range,
                                                       double range_noise, float* xyz_out, float* t_out) {
  const City* c = (const City*)h;
  const size_t n = (size_t)rings * (size_t)azimuths;
  double* rng = (double*)malloc(sizeof(double) * n);
  double* col = (double*)malloc(sizeof(double) * 12 * (size_t)azimuths);  // per column: R (9) + origin (3), world frame
  const double deg = 3.14159265358979323846 / 180.0;
  for (int j = 0; j < azimuths; j++) {
    const double tj = ((double)j / (double)azimuths - 0.5) * sweep_time;
    const double w[3] = {twist[3] * tj, twist[4] * tj, twist[5] * tj}, p[3] = {twist[0] * tj, twist[1] * tj, twist[2] * tj};
    double Rt[9];
    so3_exp(w, Rt);
    double* q = col + 12 * (size_t)j;
    for (int r = 0; r < 3; r++) {
      for (int cc = 0; cc < 3; cc++) q[3 * r + cc] = T[4 * r] * Rt[cc] + T[4 * r + 1] * Rt[3 + cc] + T[4 * r + 2] * Rt[6 + cc];
      q[9 + r] = T[4 * r] * p[0] + T[4 * r + 1] * p[1] + T[4 * r + 2] * p[2] + T[4 * r + 3];
    }
  }
  const int nt = g_threads > 0 ? g_threads : 16;
#pragma omp parallel for schedule(dynamic, 1024) num_threads(nt)
  for (long i = 0; i < (long)n; i++) {
    const int ring = (int)(i / azimuths), j = (int)(i % azimuths);
    const double el = deg * (rings > 1 ? el_top_deg + (el_bottom_deg - el_top_deg) * (double)ring / (double)(rings - 1) : el_top_deg);
    const double az = -3.14159265358979323846 + 6.283185307179586 * (double)j / (double)azimuths;
    const double dl[3] = {cos(el) * cos(az), cos(el) * sin(az), sin(el)};
    const double* q = col + 12 * (size_t)j;
    const double d[3] = {q[0] * dl[0] + q[1] * dl[1] + q[2] * dl[2], q[3] * dl[0] + q[4] * dl[1] + q[5] * dl[2],
                         q[6] * dl[0] + q[7] * dl[1] + q[8] * dl[2]};
    rng[i] = cast(c, q + 9, d, max_range);
  }
  int m = 0;
  for (size_t i = 0; i < n; i++) {
    if (!(rng[i] < max_range)) continue;
    const int ring = (int)(i / (size_t)azimuths), j = (int)(i % (size_t)azimuths);
    const double el = deg * (rings > 1 ? el_top_deg + (el_bottom_deg - el_top_deg) * (double)ring / (double)(rings - 1) : el_top_deg);
    const double az = -3.14159265358979323846 + 6.283185307179586 * (double)j / (double)azimuths;
    const double r = rng[i] + range_noise * gauss(seed, (uint64_t)i);
    xyz_out[3 * (size_t)m] = (float)(cos(el) * cos(az) * r);
    xyz_out[3 * (size_t)m + 1] = (float)(cos(el) * sin(az) * r);
    xyz_out[3 * (size_t)m + 2] = (float)(sin(el) * r);
    t_out[m] = (float)(((double)j / (double)azimuths - 0.5) * sweep_time);
    m++;
  }
  free(rng), free(col);
  return m;
}
