// ORACLE (test infrastructure only).  Thin C-ABI shim -- written for this repo, not
// copied -- that lets ctypes call the REFERENCE's own C++ core where it lies under
// /root/reference/Predator_APR/cpp_wrappers (the reference's CPython binding does not
// compile against NumPy 2).  Built by `make -C oracle ref` into oracle/_ref/ only.
//   ref_subsample_batch  -> batch_grid_subsampling   (cpp_subsampling/grid_subsampling/grid_subsampling.cpp:109)
//   ref_batch_query      -> batch_nanoflann_neighbors (cpp_neighbors/neighbors/neighbors.cpp:211)
#include <cstdlib>
#include <cstring>
#include <vector>

#include "cpp_neighbors/neighbors/neighbors.h"
#include "cpp_subsampling/grid_subsampling/grid_subsampling.h"

static std::vector<PointXYZ> to_points(const float* p, int n) {
  std::vector<PointXYZ> v(n);
  for (int i = 0; i < n; ++i) v[i] = PointXYZ(p[3 * i], p[3 * i + 1], p[3 * i + 2]);
  return v;
}

extern "C" int ref_subsample_batch(const float* pts, int n, const int* batches, int nb, float dl, int max_p,
                                   float* out_pts, int* out_batches) {
  std::vector<PointXYZ> in = to_points(pts, n), out;
  std::vector<float> f_in, f_out;
  std::vector<int> c_in, c_out, b_in(batches, batches + nb), b_out;
  batch_grid_subsampling(in, out, f_in, f_out, c_in, c_out, b_in, b_out, dl, max_p);
  for (size_t i = 0; i < out.size(); ++i) {
    out_pts[3 * i] = out[i].x;
    out_pts[3 * i + 1] = out[i].y;
    out_pts[3 * i + 2] = out[i].z;
  }
  for (int b = 0; b < nb; ++b) out_batches[b] = b_out[b];
  return (int)out.size();
}

// returns max_count; *out is malloc'ed [nq * max_count] (free with ref_free)
extern "C" int ref_batch_query(const float* q, int nq, const float* s, int ns, const int* qb, const int* sb, int nb,
                               float radius, int** out) {
  std::vector<PointXYZ> qs = to_points(q, nq), ss = to_points(s, ns);
  std::vector<int> q_b(qb, qb + nb), s_b(sb, sb + nb), idx;
  batch_nanoflann_neighbors(qs, ss, q_b, s_b, idx, radius);
  int max_count = nq > 0 ? (int)(idx.size() / nq) : 0;
  *out = (int*)std::malloc(sizeof(int) * (idx.size() + 1));
  std::memcpy(*out, idx.data(), sizeof(int) * idx.size());
  return max_count;
}

extern "C" void ref_free(void* p) { std::free(p); }
