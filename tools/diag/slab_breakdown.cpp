// tools/diag/slab_breakdown.cpp -- diagnostic: what the LDS slab of a morphology consists of (step_body.h make_layout).
//   g++ -O1 -std=c++17 -I include -o /tmp/slab_breakdown tools/diag/slab_breakdown.cpp && /tmp/slab_breakdown nb nj nq nv nu npair maxrows integrator n_int n_f64
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <cstring>
#include <cstdint>
#include "../../sgrl_amd/csrc/step_body.h"
int main(int argc, char** argv) {
  if (argc < 11) { std::printf("usage: nb nj nq nv nu npair maxrows integrator n_int n_f64\n"); return 1; }
  int32_t hdr[64] = {0};
  hdr[SGRL_H_NBODY] = atoi(argv[1]); hdr[SGRL_H_NJNT] = atoi(argv[2]); hdr[SGRL_H_NQ] = atoi(argv[3]); hdr[SGRL_H_NV] = atoi(argv[4]);
  hdr[SGRL_H_NU] = atoi(argv[5]); hdr[SGRL_H_NPAIR] = atoi(argv[6]); hdr[SGRL_H_MAX_ROWS] = atoi(argv[7]); hdr[SGRL_H_INTEGRATOR] = atoi(argv[8]);
  const int n_int = atoi(argv[9]), n_f64 = atoi(argv[10]);
  using namespace sgrl;
  Layout o; make_layout(hdr, &o, n_int, n_f64);
  const int nv = o.nv, nb = o.nb, nj = o.nj;
  std::printf("bytes %d -> %d workgroups per CU; lrows %d (na_max %d, maxrows %d), ldy %d, mfull_hbm %d\n", layout_bytes(&o),
              workgroups_per_cu(layout_bytes(&o)), o.lrows, o.na_max, o.maxrows, o.ldy, o.mfull_hbm);
  auto B = [](int d) { return d * 8; };
  std::printf("  state (qpos..ctrl)            %6d B\n", B(o.xpos));
  std::printf("  xpos, xaxis, cdof             %6d B\n", B(o.dead - o.xpos));
  std::printf("  dead zone (kinematics, inertias, contacts; reused as factor scratch) %6d B  [contacts: %d B]\n", B(o.dead_len), B(13 * o.ncon));
  std::printf("  L + dinv                      %6d B\n", B(nv * (nv + 1) / 2 + nv));
  std::printf("  qfs, qacc, vpgs               %6d B\n", B(3 * nv));
  std::printf("  Y ((lrows + 1) x ldy)         %6d B\n", B((o.lrows + 1) * o.ldy));
  std::printf("  row vectors (6 x lrows + prev)%6d B\n", B(o.misc - o.eR));
  std::printf("  misc + Mfull                  %6d B\n", B(o.model_f - o.misc));
  std::printf("  float model blob              %6d B\n", B(o.s_total - o.model_f));
  std::printf("  integer part                  %6d B  [model ints %d B]\n", ((o.i_total + 1) & ~1) * 4, n_int * 4);
  return 0;
}
