/* nn_cost_model.c -- design aid (not product, not oracle): replays the control flow of the device NN search variants
 * on the CPU to count, per scan point, the dependent round trips and load instructions it would issue, and per wave
 * (64 consecutive points) the maximum over its lanes -- the quantity a SIMT wave actually pays.
 * Build:  gcc -O2 -fopenmp -ffp-contract=off -o tools/nn_cost_model tools/nn_cost_model.c -lm
 * Input:  raw float32 files map.xyz (M*3), scan.xyz (N*3, already transformed), voxel size, cap. */
#include <stdio.h>
#include "../oracle/icp_oracle.c"

typedef struct { int rounds_p, instr_p, rounds_m, instr_m, cands_p, cands_m, probes; } cost_t;

static float gap2(float q, int c, float vs, int side) {
  const int vm = c - 1, vp = c + 1;
  const float hi_m = (float)(vm + 1) * vs, lo_p = (float)vp * vs;
  const float margin = 1.0e-6f * ((float)(c < 0 ? -c : c) + 2.0f) * vs;
  if (side == 0) { const float g = fmaxf(0.f, (q - hi_m) - margin); return g * g; }
  if (side == 2) { const float g = fmaxf(0.f, (lo_p - q) - margin); return g * g; }
  return 0.f;
}

static float scan_vox(const voxel_t* v, float qx, float qy, float qz, float best) {
  for (uint32_t j = 0; j < v->n; j++) {
    const float dx = v->xyz[3 * j] - qx, dy = v->xyz[3 * j + 1] - qy, dz = v->xyz[3 * j + 2] - qz;
    const float d2 = (dx * dx + dy * dy) + dz * dz;
    if (d2 < best) best = d2;
  }
  return best;
}

static const int CLS[3][12] = {{4, 10, 12, 14, 16, 22, -1}, {1, 3, 5, 7, 9, 11, 15, 17, 19, 21, 23, 25}, {0, 2, 6, 8, 18, 20, 24, 26, -1}};
static const int NCLS[3] = {6, 12, 8};

static void point_cost(const orc_map* m, float qx, float qy, float qz, int W, cost_t* c) {
  memset(c, 0, sizeof(*c));
  const float vs = m->p.voxel_size;
  const int32_t cx = coord2idx(m, qx), cy = coord2idx(m, qy), cz = coord2idx(m, qz);
  float g[3][3];
  for (int s = 0; s < 3; s++) { g[0][s] = gap2(qx, cx, vs, s); g[1][s] = gap2(qy, cy, vs, s); g[2][s] = gap2(qz, cz, vs, s); }
  /* ---- variant p: per voxel, 4 records per round */
  for (int variant = 0; variant < 2; variant++) {
    float best = INFINITY;
    int rounds = 0, instr = 0, cands = 0, probes = 0;
    const voxel_t* v = map_find(m, cx, cy, cz);
    rounds++; instr++; probes++;
    if (v && v->n) {
      const int r = variant == 0 ? (v->n + 3) / 4 : (v->n + W - 1) / W;
      rounds += r; instr += r * (variant == 0 ? 4 : W); cands += v->n;
      best = scan_vox(v, qx, qy, qz, best);
    }
    uint32_t mask = 0;
    for (int code = 0; code < 27; code++) {
      if (code == 13) continue;
      const float lb = (g[0][code / 9] + g[1][(code / 3) % 3]) + g[2][code % 3];
      if (!(lb * 0.9999f > best)) mask |= 1u << code;
    }
    for (int cls = 0; cls < 3; cls++) {
      int list[12], nl = 0;
      for (int k = 0; k < NCLS[cls]; k++) if (mask & (1u << CLS[cls][k])) list[nl++] = CLS[cls][k];
      int pos = 0;
      while (pos < nl) {
        if (variant == 1) { /* m re-filters by the current bound before each batch */
          int nl2 = pos;
          for (int k = pos; k < nl; k++) {
            const int code = list[k];
            const float lb = (g[0][code / 9] + g[1][(code / 3) % 3]) + g[2][code % 3];
            if (!(lb * 0.9999f > best)) list[nl2++] = code;
          }
          nl = nl2;
          if (pos >= nl) break;
        }
        const int nb = nl - pos < 4 ? nl - pos : 4;
        rounds++; instr += 4; probes += nb;
        int total = 0;
        float nbest = best;
        for (int k = 0; k < nb; k++) {
          const int code = list[pos + k];
          const float lb = (g[0][code / 9] + g[1][(code / 3) % 3]) + g[2][code % 3];
          const voxel_t* w = map_find(m, cx - 1 + code / 9, cy - 1 + (code / 3) % 3, cz - 1 + code % 3);
          if (variant == 0) {
            if (lb * 0.9999f > best) continue;
            if (w && w->n) { const int r = (w->n + 3) / 4; rounds += r; instr += 4 * r; cands += w->n; best = scan_vox(w, qx, qy, qz, best); }
          } else if (w && w->n) { total += w->n; nbest = scan_vox(w, qx, qy, qz, nbest); }
        }
        if (variant == 1 && total) { const int r = (total + W - 1) / W; rounds += r; instr += r * W; cands += total; best = nbest; }
        pos += nb;
      }
    }
    if (variant == 0) { c->rounds_p = rounds; c->instr_p = instr; c->cands_p = cands; c->probes = probes; }
    else { c->rounds_m = rounds; c->instr_m = instr; c->cands_m = cands; }
  }
}

/* generic merged-scan variant: PB probes per batch, C candidates per round trip */
static int rounds_generic(const orc_map* m, float qx, float qy, float qz, int PB, int C, int spec_faces, int* cands_out) {
  const float vs = m->p.voxel_size;
  const int32_t cx = coord2idx(m, qx), cy = coord2idx(m, qy), cz = coord2idx(m, qz);
  float g[3][3];
  for (int s = 0; s < 3; s++) { g[0][s] = gap2(qx, cx, vs, s); g[1][s] = gap2(qy, cy, vs, s); g[2][s] = gap2(qz, cz, vs, s); }
  float best = INFINITY;
  int rounds = 1, cands = 0; /* centre probe (+ speculative face probes in the same round trip) */
  const voxel_t* v = map_find(m, cx, cy, cz);
  if (v && v->n) { rounds += (v->n + C - 1) / C; cands += v->n; best = scan_vox(v, qx, qy, qz, best); }
  for (int cls = 0; cls < 3; cls++) {
    uint32_t todo = 0;
    for (int k = 0; k < NCLS[cls]; k++) todo |= 1u << CLS[cls][k];
    for (;;) {
      int list[12], nl = 0;
      for (int k = 0; k < NCLS[cls]; k++) {
        const int code = CLS[cls][k];
        if (!(todo & (1u << code))) continue;
        const float lb = (g[0][code / 9] + g[1][(code / 3) % 3]) + g[2][code % 3];
        if (!(lb * 0.9999f > best)) list[nl++] = code;
      }
      if (!nl) break;
      const int nb = nl < PB ? nl : PB;
      if (!(spec_faces && cls == 0)) rounds++; /* probe round trip (faces were probed with the centre) */
      int total = 0;
      float nbest = best;
      for (int k = 0; k < nb; k++) {
        const int code = list[k];
        todo &= ~(1u << code);
        const voxel_t* w = map_find(m, cx - 1 + code / 9, cy - 1 + (code / 3) % 3, cz - 1 + code % 3);
        if (w && w->n) { total += w->n; nbest = scan_vox(w, qx, qy, qz, nbest); }
      }
      if (total) { rounds += (total + C - 1) / C; cands += total; best = nbest; }
    }
  }
  *cands_out = cands;
  return rounds;
}

static float* read_f32(const char* path, size_t* n) {
  FILE* f = fopen(path, "rb");
  if (!f) { perror(path); exit(1); }
  fseek(f, 0, SEEK_END); const long b = ftell(f); fseek(f, 0, SEEK_SET);
  float* a = (float*)malloc(b);
  if (fread(a, 1, b, f) != (size_t)b) exit(1);
  fclose(f);
  *n = b / 4;
  return a;
}

int main(int argc, char** argv) {
  if (argc < 6) { fprintf(stderr, "usage: %s map.f32 scan.f32 voxel_size cap W [order.u32]\n", argv[0]); return 1; }
  size_t nm, ns;
  float* mp = read_f32(argv[1], &nm); nm /= 3;
  float* sp = read_f32(argv[2], &ns); ns /= 3;
  orc_map_params p = {(float)atof(argv[3]), (uint32_t)atoi(argv[4]), 0, 0.f, 0.f, 4};
  const int W = atoi(argv[5]);
  orc_map* m = orc_map_create(&p);
  float *x = malloc(nm * 4), *y = malloc(nm * 4), *z = malloc(nm * 4);
  for (size_t i = 0; i < nm; i++) { x[i] = mp[3 * i]; y[i] = mp[3 * i + 1]; z[i] = mp[3 * i + 2]; }
  orc_map_insert(m, x, y, z, nm);
  uint32_t* order = NULL;
  if (argc > 6) { size_t no; order = (uint32_t*)read_f32(argv[6], &no); }
  cost_t* c = (cost_t*)malloc(ns * sizeof(cost_t));
#pragma omp parallel for schedule(dynamic, 1024)
  for (size_t i = 0; i < ns; i++) {
    const size_t s = order ? order[i] : i;
    point_cost(m, sp[3 * s], sp[3 * s + 1], sp[3 * s + 2], W, &c[i]);
  }
  double mr_p = 0, mi_p = 0, mr_m = 0, mi_m = 0, mc_p = 0, mc_m = 0, mpb = 0;
  for (size_t i = 0; i < ns; i++) { mr_p += c[i].rounds_p; mi_p += c[i].instr_p; mr_m += c[i].rounds_m; mi_m += c[i].instr_m; mc_p += c[i].cands_p; mc_m += c[i].cands_m; mpb += c[i].probes; }
  printf("per point (mean): p rounds %.1f instr %.1f cands %.1f probes %.1f | m rounds %.1f instr %.1f cands %.1f\n", mr_p / ns, mi_p / ns, mc_p / ns, mpb / ns, mr_m / ns, mi_m / ns, mc_m / ns);
  const size_t nw = (ns + 63) / 64;
  double wr_p = 0, wi_p = 0, wr_m = 0, wi_m = 0; int tr_p = 0, tr_m = 0, ti_p = 0, ti_m = 0;
  for (size_t w = 0; w < nw; w++) {
    int a = 0, b = 0, d = 0, e = 0;
    for (size_t i = w * 64; i < ns && i < (w + 1) * 64; i++) {
      if (c[i].rounds_p > a) a = c[i].rounds_p;
      if (c[i].instr_p > b) b = c[i].instr_p;
      if (c[i].rounds_m > d) d = c[i].rounds_m;
      if (c[i].instr_m > e) e = c[i].instr_m;
    }
    wr_p += a; wi_p += b; wr_m += d; wi_m += e;
    if (a > tr_p) tr_p = a;
    if (b > ti_p) ti_p = b;
    if (d > tr_m) tr_m = d;
    if (e > ti_m) ti_m = e;
  }
  printf("per wave  (max over lanes, mean over waves | worst wave): p rounds %.1f | %d  instr %.1f | %d ;  m rounds %.1f | %d  instr %.1f | %d\n",
         wr_p / nw, tr_p, wi_p / nw, ti_p, wr_m / nw, tr_m, wi_m / nw, ti_m);
  /* generic variants */
  const int cfg[][4] = {{4, 20, 0, 16}, {8, 20, 0, 16}, {8, 40, 0, 16}, {6, 20, 1, 16}, {8, 40, 1, 16}, {12, 40, 1, 16}, {12, 80, 1, 8}, {26, 80, 1, 4}};
  for (size_t k = 0; k < sizeof(cfg) / sizeof(cfg[0]); k++) {
    int* r = (int*)malloc(ns * sizeof(int));
    double mc = 0;
#pragma omp parallel for schedule(dynamic, 1024) reduction(+ : mc)
    for (size_t i = 0; i < ns; i++) {
      const size_t sidx = order ? order[i] : i;
      int cd;
      r[i] = rounds_generic(m, sp[3 * sidx], sp[3 * sidx + 1], sp[3 * sidx + 2], cfg[k][0], cfg[k][1], cfg[k][2], &cd);
      mc += cd;
    }
    if (k == 0 && getenv("DUMP_ROUNDS")) { FILE* f = fopen(getenv("DUMP_ROUNDS"), "wb"); fwrite(r, sizeof(int), ns, f); fclose(f); }
    const int ppw = cfg[k][3];
    double mean = 0, wmean = 0; int worst = 0; size_t nwv = (ns + ppw - 1) / ppw;
    int hist[64] = {0};
    for (size_t i = 0; i < ns; i++) mean += r[i];
    for (size_t w = 0; w < nwv; w++) {
      int a = 0;
      for (size_t i = w * ppw; i < ns && i < (w + 1) * ppw; i++) if (r[i] > a) a = r[i];
      wmean += a; if (a > worst) worst = a;
      hist[a < 63 ? a : 63]++;
    }
    printf("PB=%2d C=%2d spec_faces=%d points/wave=%2d: rounds/point %.2f  cands/point %.1f  wave-max mean %.2f  worst wave %d   waves with >=12 rounds: ", cfg[k][0], cfg[k][1], cfg[k][2], ppw, mean / ns, mc / ns, wmean / nwv, worst);
    int cnt = 0; for (int h = 12; h < 64; h++) cnt += hist[h];
    printf("%d of %zu\n", cnt, nwv);
    free(r);
  }
  return 0;
}
