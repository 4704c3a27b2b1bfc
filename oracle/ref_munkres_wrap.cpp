// ref_munkres_wrap.cpp -- TEST INFRASTRUCTURE.  C entry point around the REFERENCE's own assignment
// solver (auv_ekf_slam/utils/munkres/munkres.h, the Munkres<double> that
// auv_ekf_slam/src/ekf_slam_core.cpp:298-312 calls).  The reference sources are compiled from where
// they lie under /root/reference (oracle/Makefile, target _ref/libref_munkres.so); nothing of them is
// copied into this repository.  Used only by tests to pin oracle/mcl_oracle.c:orc_assign_dense.
#include <cstddef>

#include "munkres.h"

extern "C" int ref_munkres_solve(int rows, int cols, const double* cost, int* row_of_col) {
  // same call sequence as ekf_slam_core.cpp:298-312: fill Matrix<double>, solve in place, the
  // assignments are the entries left at 0
  Matrix<double> m(rows, cols);
  for (int r = 0; r < rows; ++r)
    for (int c = 0; c < cols; ++c) m(r, c) = cost[(size_t)r * cols + c];
  Munkres<double> solver;
  solver.solve(m);
  for (int c = 0; c < cols; ++c) {
    row_of_col[c] = -1;
    for (int r = 0; r < rows; ++r)
      if (m(r, c) == 0) {
        row_of_col[c] = r;
        break;
      }
  }
  return 0;
}
