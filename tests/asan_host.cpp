// Host-only exercise of the C ABI glue (mvae_create / tables / validation / destroy / misuse before bind) for an
// AddressSanitizer + UBSan build of csrc/runtime.cpp (the race / memory-error stand-in that is
// possible here -- GPU AddressSanitizer is not available on this pool).  No HIP call is made: nothing needs a GPU.
#include <algorithm>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>
#include "../include/mvae_hip.h"

static mvae_config nb_config(int h, int w, int levels, int max_batch) {
  mvae_config c;
  memset(&c, 0, sizeof(c));
  c.abi_version = MVAE_ABI_VERSION;
  c.input_h = h; c.input_w = w; c.input_c = 3; c.levels = levels;
  for (int i = 0; i < levels; ++i) c.z_dims[i] = 16;
  const int f[5] = {64, 64, 64, 64, 32}, k[5] = {5, 3, 3, 1, 1}, s[5] = {2, 1, 1, 1, 1};
  c.enc_n = c.dec_n = 5;
  for (int i = 0; i < 5; ++i) {
    c.enc_filters[i] = c.dec_filters[i] = f[i];
    c.enc_kh[i] = c.enc_kw[i] = c.dec_kh[i] = c.dec_kw[i] = k[i];
    c.enc_sh[i] = c.enc_sw[i] = c.dec_sh[i] = c.dec_sw[i] = s[i];
  }
  c.min_value = 0.f; c.max_value = 255.f; c.sample_std = 0.01f; c.max_batch = max_batch;
  return c;
}

#define CHECK(cond) do { if (!(cond)) { fprintf(stderr, "FAILED %s:%d: %s\n", __FILE__, __LINE__, #cond); return 1; } } while (0)

int main() {
  int created = 0;
  for (int dt = 0; dt < 2; ++dt) {
    const int sizes[3][3] = {{32, 32, 3}, {64, 64, 5}, {256, 256, 7}};
    for (auto& sz : sizes) {
      mvae_config c = nb_config(sz[0], sz[1], sz[2], 8);
      c.act_dtype = dt;
      mvae_handle* h = nullptr;
      CHECK(mvae_create(&c, &h) == MVAE_OK && h);
      ++created;
      const int64_t np = mvae_param_count(h), ns = mvae_state_count(h);
      CHECK(np == 140 * sz[2] && ns > 0 && mvae_param_elems(h) > 0 && mvae_workspace_bytes(h) > 0);
      CHECK(mvae_reduce_elems(h) == mvae_param_elems(h) + mvae_state_elems(h) + (mvae_reduce_elems(h) - mvae_metrics_offset(h)));
      std::vector<std::pair<int64_t, int64_t>> slots;   // the Dense weights lead the arena (mvae_reduce_split): sort by offset
      for (int64_t i = 0; i < np; ++i) {
        char name[MVAE_NAME_CAP];
        int64_t shape[4], off; int32_t nd, reg;
        CHECK(mvae_param_info(h, i, name, sizeof(name), shape, &nd, &off, &reg) == MVAE_OK);
        int64_t n = 1; for (int k = 0; k < nd; ++k) n *= shape[k];
        CHECK(off >= 0 && off % 64 == 0 && n > 0 && strlen(name) > 0 && reg >= 0 && reg <= 2);
        slots.push_back({off, n});
        char tiny[4];                                            // truncation must stay inside the caller's buffer
        CHECK(mvae_param_info(h, i, tiny, sizeof(tiny), nullptr, nullptr, nullptr, nullptr) == MVAE_OK && strlen(tiny) <= 3);
      }
      std::sort(slots.begin(), slots.end());
      int64_t prev_end = 0;
      for (auto& sl : slots) { CHECK(sl.first >= prev_end); prev_end = sl.first + sl.second; }
      CHECK(prev_end <= mvae_param_elems(h));
      const int64_t split = mvae_reduce_split(h);               // a slot boundary: no tensor straddles the two messages
      CHECK(split >= 0 && split % 64 == 0 && split <= mvae_param_elems(h));
      for (auto& sl : slots) CHECK(sl.first + sl.second <= split || sl.first >= split);
      CHECK(mvae_param_info(h, np, nullptr, 0, nullptr, nullptr, nullptr, nullptr) == MVAE_E_INVALID);
      for (int64_t i = 0; i < ns; ++i) {
        char name[MVAE_NAME_CAP]; int64_t n, off;
        CHECK(mvae_state_info(h, i, name, sizeof(name), &n, &off) == MVAE_OK && n > 0);
      }
      CHECK(mvae_state_info(h, -1, nullptr, 0, nullptr, nullptr) == MVAE_E_INVALID);
      for (int s = 0; s < sz[2]; ++s) { int d = mvae_scale_dtype(h, s); CHECK(d == MVAE_ACT_F32 || d == (dt ? MVAE_ACT_BF16 : MVAE_ACT_F32)); }
      CHECK(mvae_scale_dtype(h, sz[2]) == MVAE_E_INVALID);
      // misuse before bind: errors, not crashes
      mvae_step_io io; memset(&io, 0, sizeof(io)); io.batch = 4; io.training = 1;
      CHECK(mvae_forward(h, &io, nullptr) == MVAE_E_STATE);
      CHECK(mvae_backward(h, 1.f, 1.f, nullptr) == MVAE_E_STATE);
      CHECK(mvae_apply_adagrad(h, 1e-3f, 1.f, 1.f, nullptr) == MVAE_E_STATE);
      CHECK(mvae_train_step(h, &io, 1.f, 1.f, 1e-3f, 1.f, nullptr) == MVAE_E_STATE);
      float* p = nullptr; int64_t n = 0; int32_t dtp = -1;
      CHECK(mvae_tensor_lookup(h, "recon", &p, &n) == MVAE_OK && p == nullptr && n == (int64_t)sz[0] * sz[1] * 3);
      CHECK(mvae_tensor_lookup2(h, "enc0.b0.mn.t0", (void**)&p, &n, &dtp) == MVAE_OK && dtp == (dt ? MVAE_ACT_BF16 : MVAE_ACT_F32));
      CHECK(mvae_tensor_lookup(h, "no such tensor", &p, &n) == MVAE_E_INVALID);
      CHECK(strlen(mvae_last_error(h)) > 0);
      mvae_destroy(h);
    }
  }
  // constructor validation (multiscale_vae.py:34-38, layer_blocks.py:918-927): every rejection leaves no handle behind
  {
    mvae_handle* h = nullptr;
    mvae_config c = nb_config(32, 32, 3, 8);
    c.z_dims[1] = 0;                     CHECK(mvae_create(&c, &h) == MVAE_E_INVALID && !h && strstr(mvae_last_error(nullptr), "z_dims"));
    c = nb_config(32, 32, 1, 8);         CHECK(mvae_create(&c, &h) == MVAE_E_INVALID && !h);
    c = nb_config(30, 32, 3, 8);         CHECK(mvae_create(&c, &h) == MVAE_E_INVALID && !h);
    c = nb_config(32, 32, 3, 0);         CHECK(mvae_create(&c, &h) == MVAE_E_INVALID && !h);
    c = nb_config(32, 32, 3, 8); c.enc_filters[2] = 0; CHECK(mvae_create(&c, &h) == MVAE_E_INVALID && !h);
    c = nb_config(32, 32, 3, 8); c.abi_version = 1;    CHECK(mvae_create(&c, &h) == MVAE_E_INVALID && !h);
    c = nb_config(32, 32, 3, 8); c.act_dtype = 7;      CHECK(mvae_create(&c, &h) == MVAE_E_INVALID && !h);
    c = nb_config(32, 32, 3, 8); c.dec_sh[0] = 1;      CHECK(mvae_create(&c, &h) == MVAE_E_INVALID && !h);   // decoder does not return to 32x32
    c = nb_config(32, 32, 3, 8); c.levels = MVAE_MAX_LEVELS + 1; CHECK(mvae_create(&c, &h) == MVAE_E_INVALID && !h);
    CHECK(mvae_create(nullptr, &h) == MVAE_E_INVALID);
    CHECK(mvae_param_count(nullptr) == -1 && mvae_workspace_bytes(nullptr) == -1);
    mvae_destroy(nullptr);
  }
  printf("asan_host: %d handles created and destroyed, validation paths exercised\n", created);
  return 0;
}
