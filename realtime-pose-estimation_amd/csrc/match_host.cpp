// Host-side grouping of the top-K candidates into people: match_by_tag
// (rtpe/third_party/group.py:26-97 of the reference) with the Kuhn-Munkres
// assignment that the reference gets from the PyPI package `munkres`
// (group.py:14,19-23).  Sequential by construction (joint j depends on the
// people built from joints < j), tiny (<= 30 x 30 per joint), so it stays on the
// host - in C++ instead of numpy + pure-Python Munkres.
//
// The arithmetic follows numpy's: rows are float64 (int64 locations, float32
// values and tags promoted), tag means are float32 with numpy's reduction
// order (8-way pairwise over a contiguous run when D == 1, sequential
// otherwise), distances are float64 sqrt of a float64 sum of squares.
// The assignment follows the package's published procedure including its scan
// orders (see oracle/hungarian_ref.py for the statement that is tested
// against this file).
#include <math.h>
#include <stdint.h>
#include <string.h>

#include <algorithm>
#include <thread>
#include <vector>

#include "rtpe_hip.h"

namespace rtpe {
void set_error(const char* fmt, ...);
}

namespace {

// ---- Kuhn-Munkres, munkres-1.1.x procedure --------------------------------
struct Munkres {
  int n;
  std::vector<double> C;
  std::vector<char> row_cov, col_cov, mark;   // mark: 1 star, 2 prime

  double& c(int i, int j) { return C[(size_t)i * n + j]; }
  char& mk(int i, int j) { return mark[(size_t)i * n + j]; }

  bool find_zero(int i0, int j0, int* ri, int* rj) {
    int i = i0;
    while (true) {
      int hit = -1, j = j0;
      while (true) {
        if (c(i, j) == 0.0 && !row_cov[i] && !col_cov[j]) hit = j;   // keeps the LAST one of the row
        j = (j + 1) % n;
        if (j == j0) break;
      }
      if (hit >= 0) { *ri = i; *rj = hit; return true; }
      i = (i + 1) % n;
      if (i == i0) return false;
    }
  }

  // cost: nr x nc row-major; pairs written as (row, col) in row-major order
  void compute(const double* cost, int nr, int nc, std::vector<std::pair<int, int>>* pairs) {
    pairs->clear();
    if (nr <= 0 || nc <= 0) return;
    n = std::max(nr, nc);
    C.assign((size_t)n * n, 0.0);
    for (int i = 0; i < nr; ++i)
      for (int j = 0; j < nc; ++j) c(i, j) = cost[(size_t)i * nc + j];
    row_cov.assign(n, 0);
    col_cov.assign(n, 0);
    mark.assign((size_t)n * n, 0);
    for (int i = 0; i < n; ++i) {                       // step 1
      double m = c(i, 0);
      for (int j = 1; j < n; ++j) m = std::min(m, c(i, j));
      for (int j = 0; j < n; ++j) c(i, j) -= m;
    }
    for (int i = 0; i < n; ++i)                         // step 2
      for (int j = 0; j < n; ++j)
        if (c(i, j) == 0.0 && !col_cov[j] && !row_cov[i]) {
          mk(i, j) = 1; col_cov[j] = 1; row_cov[i] = 1;
          break;
        }
    std::fill(row_cov.begin(), row_cov.end(), 0);
    std::fill(col_cov.begin(), col_cov.end(), 0);
    int step = 3, z0r = 0, z0c = 0;
    std::vector<std::pair<int, int>> path;
    while (true) {
      if (step == 3) {
        int count = 0;
        for (int i = 0; i < n; ++i)
          for (int j = 0; j < n; ++j)
            if (mk(i, j) == 1 && !col_cov[j]) { col_cov[j] = 1; ++count; }
        if (count >= n) break;
        step = 4;
      } else if (step == 4) {
        int row = 0, col = 0;
        while (true) {
          int r, q;
          if (!find_zero(row, col, &r, &q)) { step = 6; break; }
          row = r; col = q;
          mk(row, col) = 2;
          int star = -1;
          for (int j = 0; j < n; ++j) if (mk(row, j) == 1) { star = j; break; }
          if (star >= 0) {
            col = star;
            row_cov[row] = 1;
            col_cov[col] = 0;
          } else {
            z0r = row; z0c = col; step = 5;
            break;
          }
        }
      } else if (step == 5) {
        path.clear();
        path.push_back({z0r, z0c});
        while (true) {
          int r = -1;
          for (int i = 0; i < n; ++i) if (mk(i, path.back().second) == 1) { r = i; break; }
          if (r < 0) break;
          path.push_back({r, path.back().second});
          int q = -1;
          for (int j = 0; j < n; ++j) if (mk(r, j) == 2) { q = j; break; }
          path.push_back({r, q});
        }
        for (auto& pc : path) mk(pc.first, pc.second) = mk(pc.first, pc.second) == 1 ? 0 : 1;
        std::fill(row_cov.begin(), row_cov.end(), 0);
        std::fill(col_cov.begin(), col_cov.end(), 0);
        for (auto& v : mark) if (v == 2) v = 0;
        step = 3;
      } else {                                            // step 6
        double m = INFINITY;
        for (int i = 0; i < n; ++i)
          if (!row_cov[i])
            for (int j = 0; j < n; ++j)
              if (!col_cov[j]) m = std::min(m, c(i, j));
        for (int i = 0; i < n; ++i)
          for (int j = 0; j < n; ++j) {
            if (row_cov[i]) c(i, j) += m;
            if (!col_cov[j]) c(i, j) -= m;
          }
        step = 4;
      }
    }
    for (int i = 0; i < nr; ++i)
      for (int j = 0; j < nc; ++j)
        if (mk(i, j) == 1) pairs->push_back({i, j});
  }
};

float pairwise8_f32(const float* a, int n, int stride) {   // numpy contiguous float32 add.reduce
  if (n < 8) {
    float s = 0.f;
    for (int i = 0; i < n; ++i) s = s + a[(size_t)i * stride];
    return s;
  }
  float r[8];
  for (int j = 0; j < 8; ++j) r[j] = a[(size_t)j * stride];
  int i = 8;
  for (; i + 8 <= n; i += 8)
    for (int j = 0; j < 8; ++j) r[j] = r[j] + a[(size_t)(i + j) * stride];
  float s = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
  for (; i < n; ++i) s = s + a[(size_t)i * stride];
  return s;
}

double pairwise8_f64(const double* a, int n) {
  if (n < 8) {
    double s = 0.0;
    for (int i = 0; i < n; ++i) s = s + a[i];
    return s;
  }
  double r[8];
  for (int j = 0; j < 8; ++j) r[j] = a[j];
  int i = 8;
  for (; i + 8 <= n; i += 8)
    for (int j = 0; j < 8; ++j) r[j] = r[j] + a[i + j];
  double s = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
  for (; i < n; ++i) s = s + a[i];
  return s;
}

struct Person {
  float key;                     // float value of tag[0] of the founding joint (the dict key)
  std::vector<double> rows;      // J x (3+D)
  std::vector<float> tags;       // n_tags x D (tag_dict[key])
  int n_tags;
};

}  // namespace

namespace {

// one image; appends the people rows (J x (3+D) float32 each) to `out`
void match_image(const float* tag_k, const int32_t* ind_k, const float* val_k, int32_t J, int32_t K, int32_t D,
                 int32_t w, int32_t max_num_people, double detection_threshold, double tag_threshold,
                 int32_t use_detection_val, int32_t ignore_too_much, std::vector<float>* out, int* n_found) {
  const int R = 3 + D;
  // numpy compares float64(val) with the Python float thresholds: keep them double
  const double det_thr = detection_threshold;
  const double tag_thr = tag_threshold;
  std::vector<Person> people;     // insertion-ordered dict
  Munkres solver;
  std::vector<std::pair<int, int>> pairs;
  std::vector<double> rows, dist, cost, sq(D > 0 ? D : 1);
  std::vector<float> tags, centres;
  std::vector<int> keep;

  auto find_person = [&](float key) -> int {
    for (size_t i = 0; i < people.size(); ++i)
      if (people[i].key == key) return (int)i;   // dict semantics: == on the float value
    return -1;
  };
  auto put = [&](int j, const double* row, const float* tag) {
    int pi = find_person(tag[0]);
    if (pi < 0) {
      Person p;
      p.key = tag[0];
      p.rows.assign((size_t)J * R, 0.0);
      p.n_tags = 0;
      people.push_back(std::move(p));
      pi = (int)people.size() - 1;
    }
    Person& p = people[pi];
    memcpy(&p.rows[(size_t)j * R], row, sizeof(double) * R);
    p.tags.assign(tag, tag + D);                  // tag_dict[key] = [tag]
    p.n_tags = 1;
  };

  for (int j = 0; j < J; ++j) {
    keep.clear();
    for (int k = 0; k < K; ++k)
      if ((double)val_k[(size_t)j * K + k] > det_thr) keep.push_back(k);
    const int A = (int)keep.size();
    if (A == 0) continue;
    rows.assign((size_t)A * R, 0.0);
    tags.assign((size_t)A * D, 0.f);
    for (int a = 0; a < A; ++a) {
      const int k = keep[a];
      const int ind = ind_k[(size_t)j * K + k];
      rows[(size_t)a * R + 0] = (double)(ind % w);
      rows[(size_t)a * R + 1] = (double)(ind / w);
      rows[(size_t)a * R + 2] = (double)val_k[(size_t)j * K + k];
      for (int d = 0; d < D; ++d) {
        const float t = tag_k[((size_t)j * K + k) * D + d];
        rows[(size_t)a * R + 3 + d] = (double)t;
        tags[(size_t)a * D + d] = t;
      }
    }
    if (j == 0 || people.empty()) {
      for (int a = 0; a < A; ++a) put(j, &rows[(size_t)a * R], &tags[(size_t)a * D]);
      continue;
    }
    const int G = std::min((int)people.size(), (int)max_num_people);
    if (ignore_too_much && G == max_num_people) continue;
    centres.assign((size_t)G * D, 0.f);
    for (int g = 0; g < G; ++g) {
      const Person& p = people[g];
      for (int d = 0; d < D; ++d) {
        float s;
        if (D == 1) {
          s = pairwise8_f32(p.tags.data(), p.n_tags, 1);
        } else {
          s = 0.f;
          for (int t = 0; t < p.n_tags; ++t) s = s + p.tags[(size_t)t * D + d];
        }
        centres[(size_t)g * D + d] = s / (float)p.n_tags;
      }
    }
    const int NC = A > G ? A : G;
    dist.assign((size_t)A * G, 0.0);
    cost.assign((size_t)A * NC, 1e10);
    for (int a = 0; a < A; ++a)
      for (int g = 0; g < G; ++g) {
        for (int d = 0; d < D; ++d) {
          const double df = rows[(size_t)a * R + 3 + d] - (double)centres[(size_t)g * D + d];
          sq[d] = df * df;
        }
        const double dd = sqrt(pairwise8_f64(sq.data(), D));
        dist[(size_t)a * G + g] = dd;
        cost[(size_t)a * NC + g] = use_detection_val ? nearbyint(dd) * 100.0 - rows[(size_t)a * R + 2] : dd;
      }
    solver.compute(cost.data(), A, NC, &pairs);
    for (auto& rc : pairs) {
      const int r = rc.first, q = rc.second;
      if (r < A && q < G && dist[(size_t)r * G + q] < tag_thr) {
        Person& p = people[q];
        memcpy(&p.rows[(size_t)j * R], &rows[(size_t)r * R], sizeof(double) * R);
        p.tags.insert(p.tags.end(), &tags[(size_t)r * D], &tags[(size_t)r * D] + D);
        p.n_tags += 1;
      } else {
        put(j, &rows[(size_t)r * R], &tags[(size_t)r * D]);
      }
    }
  }
  *n_found = (int)people.size();
  out->reserve(out->size() + people.size() * (size_t)J * R);
  for (const Person& p : people)
    for (int i = 0; i < J * R; ++i) out->push_back((float)p.rows[i]);
}

}  // namespace

extern "C" int rtpe_match_by_tag(const float* tag_k, const int32_t* ind_k, const float* val_k, int32_t J,
                                 int32_t K, int32_t D, int32_t w, int32_t max_num_people,
                                 double detection_threshold, double tag_threshold, int32_t use_detection_val,
                                 int32_t ignore_too_much, float* ans, int32_t max_people_out,
                                 int32_t* n_people) {
  if (!tag_k || !ind_k || !val_k || !n_people || J <= 0 || K <= 0 || D <= 0 || w <= 0 ||
      (max_people_out > 0 && !ans)) {
    rtpe::set_error("match_by_tag: bad argument");
    return RTPE_E_INVALID;
  }
  std::vector<float> rows;
  int n = 0;
  match_image(tag_k, ind_k, val_k, J, K, D, w, max_num_people, detection_threshold, tag_threshold,
              use_detection_val, ignore_too_much, &rows, &n);
  *n_people = n;
  const size_t keep = (size_t)std::min(n, (int)max_people_out) * J * (3 + D);
  if (keep) memcpy(ans, rows.data(), keep * sizeof(float));
  return RTPE_OK;
}

extern "C" int rtpe_match_by_tag_batch(const float* tag_k, const int32_t* ind_k, const float* val_k, int32_t N,
                                       int32_t J, int32_t K, int32_t D, int32_t w, int32_t max_num_people,
                                       double detection_threshold, double tag_threshold,
                                       int32_t use_detection_val, int32_t ignore_too_much, float* ans,
                                       int32_t max_people_total, int32_t* person_img, int32_t* counts,
                                       int32_t n_threads) {
  if (!tag_k || !ind_k || !val_k || !counts || N <= 0 || J <= 0 || K <= 0 || D <= 0 || w <= 0 ||
      (max_people_total > 0 && (!ans || !person_img))) {
    rtpe::set_error("match_by_tag_batch: bad argument");
    return RTPE_E_INVALID;
  }
  std::vector<std::vector<float>> rows(N);
  std::vector<int> found(N, 0);
  auto work = [&](int t, int nt) {
    for (int n = t; n < N; n += nt)
      match_image(tag_k + (size_t)n * J * K * D, ind_k + (size_t)n * J * K, val_k + (size_t)n * J * K, J, K, D, w,
                  max_num_people, detection_threshold, tag_threshold, use_detection_val, ignore_too_much,
                  &rows[n], &found[n]);
  };
  int nt = n_threads < 1 ? 1 : (n_threads > N ? N : n_threads);
  if (nt == 1) {
    work(0, 1);
  } else {
    std::vector<std::thread> th;
    for (int t = 0; t < nt; ++t) th.emplace_back(work, t, nt);
    for (auto& x : th) x.join();
  }
  const size_t R = (size_t)J * (3 + D);
  size_t o = 0;
  for (int n = 0; n < N; ++n) {
    counts[n] = found[n];
    for (int q = 0; q < found[n] && (int)o < max_people_total; ++q, ++o) {
      memcpy(ans + o * R, rows[n].data() + (size_t)q * R, R * sizeof(float));
      person_img[o] = n;
    }
  }
  return RTPE_OK;
}

// Munkres().compute(cost) of the PyPI package used at group.py:19-23
// (py_max_match).  cost: nr x nc row-major doubles; pairs: out, 2*min(nr,nc)
// ints (row, col); *n_pairs = number of pairs written.  Host function.
extern "C" int rtpe_munkres(const double* cost, int32_t nr, int32_t nc, int32_t* pairs, int32_t* n_pairs) {
  if (!cost || !pairs || !n_pairs || nr <= 0 || nc <= 0) {
    rtpe::set_error("munkres: bad argument");
    return RTPE_E_INVALID;
  }
  Munkres solver;
  std::vector<std::pair<int, int>> out;
  solver.compute(cost, nr, nc, &out);
  *n_pairs = (int)out.size();
  for (size_t i = 0; i < out.size(); ++i) { pairs[2 * i] = out[i].first; pairs[2 * i + 1] = out[i].second; }
  return RTPE_OK;
}
