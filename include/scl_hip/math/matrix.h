// include/scl_hip/math/matrix.h -- scl::math::Matrix<T> (include/scl/math/matrix.h:52-968).
//
// Row-major host storage like the reference.  The data-parallel members -- multiply(Matrix),
// multiply(Vector), the entry-wise family, vandermonde -- run on the GPU behind the C ABI (scl_hip_matmul,
// scl_hip_ew, scl_hip_vandermonde) from hip::hostThreshold() multiply-adds / entries on, and as the reference's
// loops over FF's operators below it (hip/device.h); hyperInvertible builds its rows through computeLagrangeBasis.  invert() is a small dense Gauss-Jordan on scalars and stays host work, as in the
// reference ("invert/solveLinearSystem host-side only").
#ifndef SCL_HIP_MATH_MATRIX_H
#define SCL_HIP_MATH_MATRIX_H

#include <iomanip>
#include <sstream>
#include <stdexcept>
#include <string>
#include <vector>

#include "lagrange.h"
#include "vector.h"

namespace scl::math {

template <typename ELEMENT>
class Matrix {
  static constexpr bool DEV = OnDevice<ELEMENT>;  // else: the reference's loops on the host (user-defined fields)

 public:
  using ValueType = ELEMENT;

  static Matrix random(std::size_t n, std::size_t m, util::PRG& prg) {
    return Matrix(n, m, Vector<ELEMENT>::random(n * m, prg).toStlVector());
  }

  /// V(i, j) = xs[i]^j (matrix.h:444-460)
  static Matrix vandermonde(std::size_t n, std::size_t m, const Vector<ELEMENT>& xs) {
    if (xs.size() != n) hip::detail::raise(SCL_ERR_VANDERMONDE_XS);
    if (hip::onHost<ELEMENT>(n * m)) {  // V(i,0) = 1, V(i,j) = V(i,j-1) * xs[i]
      Matrix v(n, m);
      for (std::size_t i = 0; i < n; ++i) {
        v(i, 0) = ELEMENT(1);
        for (std::size_t j = 1; j < m; ++j) v(i, j) = v(i, j - 1) * xs[i];
      }
      return v;
    }
    if constexpr (DEV) {
      constexpr std::size_t L = hip::limbsOf<ELEMENT>();
      hip::DeviceVector<ELEMENT> v(n * m);
      std::vector<std::uint64_t> x(n * L + 1);
      for (std::size_t i = 0; i < n; ++i) xs[i].toLimbs(x.data() + i * L);
      hip::check(scl_hip_vandermonde(ELEMENT::Field::TAG, v.data(), n, m, x.data(), nullptr));
      return Matrix(n, m, v.toHost());
    } else {
      hip::detail::unreachable();
    }
  }
  /// default nodes 1..n (matrix.h:102-104)
  static Matrix vandermonde(std::size_t n, std::size_t m) { return vandermonde(n, m, Vector<ELEMENT>::range(1, n + 1)); }

  /// row i = Lagrange basis of nodes 1..m evaluated at -i (matrix.h:462-475)
  static Matrix hyperInvertible(std::size_t n, std::size_t m) {
    Matrix him(n, m);
    const auto vs = Vector<ELEMENT>::range(1, m + 1);
    for (std::size_t i = 0; i < n; ++i) {
      const auto r = computeLagrangeBasis(vs, -static_cast<int>(i));
      for (std::size_t j = 0; j < m; ++j) him(i, j) = r[j];
    }
    return him;
  }

  static Matrix fromVector(std::size_t n, std::size_t m, const std::vector<ELEMENT>& vec) {
    if (vec.size() != n * m) throw std::invalid_argument("invalid dimensions");
    return Matrix(n, m, vec);
  }

  static Matrix identity(std::size_t n) {
    Matrix id(n);
    for (std::size_t i = 0; i < n; ++i) id(i, i) = ELEMENT(1);
    return id;
  }

  Matrix() : m_rows(0), m_cols(0) {}
  explicit Matrix(std::size_t n, std::size_t m) : m_rows(n), m_cols(m), m_values(n * m) {
    if (n == 0 || m == 0) throw std::invalid_argument("n or m cannot be 0");
  }
  explicit Matrix(std::size_t n) : Matrix(n, n) {}

  std::size_t rows() const { return m_rows; }
  std::size_t cols() const { return m_cols; }
  ELEMENT& operator()(std::size_t r, std::size_t c) { return m_values[r * m_cols + c]; }
  ELEMENT operator()(std::size_t r, std::size_t c) const { return m_values[r * m_cols + c]; }

  Matrix add(const Matrix& o) const { return entrywise(o, &Vector<ELEMENT>::add); }
  Matrix subtract(const Matrix& o) const { return entrywise(o, &Vector<ELEMENT>::subtract); }
  Matrix multiplyEntryWise(const Matrix& o) const { return entrywise(o, &Vector<ELEMENT>::multiplyEntryWise); }
  Matrix& addInPlace(const Matrix& o) { return *this = add(o); }
  Matrix& subtractInPlace(const Matrix& o) { return *this = subtract(o); }
  Matrix& multiplyEntryWiseInPlace(const Matrix& o) { return *this = multiplyEntryWise(o); }
  Matrix scalarMultiply(const ELEMENT& s) const {
    return Matrix(m_rows, m_cols, Vector<ELEMENT>(m_values).scalarMultiply(s).toStlVector());
  }
  Matrix& scalarMultiplyInPlace(const ELEMENT& s) { return *this = scalarMultiply(s); }

  /// C = this * other on the GPU (matrix.h:477-495)
  Matrix multiply(const Matrix& o) const {
    if (cols() != o.rows()) hip::detail::raise(SCL_ERR_MATMUL_DIMS);
    if (hip::onHost<ELEMENT>(rows() * cols() * o.cols())) {  // i-k-j triple loop
      Matrix r(rows(), o.cols());
      for (std::size_t i = 0; i < rows(); ++i)
        for (std::size_t k = 0; k < cols(); ++k)
          for (std::size_t j = 0; j < o.cols(); ++j) r(i, j) += (*this)(i, k) * o(k, j);
      return r;
    }
    if constexpr (DEV) {
      hip::DeviceVector<ELEMENT> a(m_values), b(o.m_values), c(rows() * o.cols());
      hip::check(scl_hip_matmul(ELEMENT::Field::TAG, c.data(), o.cols(), a.data(), cols(), b.data(), o.cols(), rows(), cols(),
                                o.cols(), nullptr));
      return Matrix(rows(), o.cols(), c.toHost());
    } else {
      hip::detail::unreachable();
    }
  }

  /// matrix-vector product (matrix.h:497-513)
  Vector<ELEMENT> multiply(const Vector<ELEMENT>& v) const {
    if (cols() != v.size()) throw std::invalid_argument("matmul: this->cols() != vec.size()");
    if (hip::onHost<ELEMENT>(rows() * cols())) {  // one innerProd per row
      std::vector<ELEMENT> r;
      r.reserve(rows());
      for (std::size_t i = 0; i < rows(); ++i)
        r.emplace_back(innerProd<ELEMENT>(m_values.begin() + static_cast<std::ptrdiff_t>(i * cols()),
                                          m_values.begin() + static_cast<std::ptrdiff_t>((i + 1) * cols()), v.begin()));
      return Vector<ELEMENT>(std::move(r));
    }
    if constexpr (DEV) {
      hip::DeviceVector<ELEMENT> a(m_values), b(v.toStlVector()), c(rows());
      hip::check(scl_hip_matmul(ELEMENT::Field::TAG, c.data(), 1, a.data(), cols(), b.data(), 1, rows(), cols(), 1, nullptr));
      return Vector<ELEMENT>(c.toHost());
    } else {
      hip::detail::unreachable();
    }
  }

  Matrix transpose() const {
    Matrix t(m_cols, m_rows);
    for (std::size_t i = 0; i < m_rows; ++i)
      for (std::size_t j = 0; j < m_cols; ++j) t(j, i) = (*this)(i, j);
    return t;
  }

  Matrix& resize(std::size_t new_rows, std::size_t new_cols) {
    if (new_rows * new_cols != m_rows * m_cols) throw std::invalid_argument("cannot resize matrix");
    m_rows = new_rows;
    m_cols = new_cols;
    return *this;
  }

  bool isSquare() const { return m_rows == m_cols; }

  bool isIdentity() const {
    if (!isSquare()) return false;
    bool ok = true;
    for (std::size_t i = 0; i < m_rows; ++i)
      for (std::size_t j = 0; j < m_cols; ++j) ok &= (*this)(i, j) == (i == j ? ELEMENT{1} : ELEMENT{0});
    return ok;
  }

  /// Gauss-Jordan inverse of a square matrix (matrix.h:830-850)
  Matrix invert() const {
    if (!isSquare()) throw std::invalid_argument("cannot invert non-square matrix");
    const std::size_t n = m_rows;
    Matrix a = *this, inv = identity(n);
    for (std::size_t col = 0; col < n; ++col) {
      std::size_t piv = col;
      while (piv < n && a(piv, col) == ELEMENT{}) ++piv;
      if (piv == n) continue;  // singular: the reference returns whatever the reduction leaves
      if (piv != col)
        for (std::size_t j = 0; j < n; ++j) {
          std::swap(a.m_values[piv * n + j], a.m_values[col * n + j]);
          std::swap(inv.m_values[piv * n + j], inv.m_values[col * n + j]);
        }
      const ELEMENT s = a(col, col).inverse();
      for (std::size_t j = 0; j < n; ++j) {
        a(col, j) *= s;
        inv(col, j) *= s;
      }
      for (std::size_t r = 0; r < n; ++r) {
        if (r == col) continue;
        const ELEMENT f = a(r, col);
        if (f == ELEMENT{}) continue;
        for (std::size_t j = 0; j < n; ++j) {
          a(r, j) -= f * a(col, j);
          inv(r, j) -= f * inv(col, j);
        }
      }
    }
    return inv;
  }

  bool equals(const Matrix& o) const {
    if (rows() != o.rows() || cols() != o.cols()) return false;
    return Vector<ELEMENT>(m_values).equals(Vector<ELEMENT>(o.m_values));
  }
  friend bool operator==(const Matrix& l, const Matrix& r) { return l.equals(r); }
  friend bool operator!=(const Matrix& l, const Matrix& r) { return !l.equals(r); }

  std::string toString() const {
    if (!(m_rows && m_cols)) return "[ EMPTY MATRIX ]";
    std::vector<std::size_t> width(m_cols, 0);
    for (std::size_t j = 0; j < m_cols; ++j)
      for (std::size_t i = 0; i < m_rows; ++i) width[j] = std::max(width[j], (*this)(i, j).toString().size());
    std::stringstream ss;
    ss << "\n";
    for (std::size_t i = 0; i < m_rows; ++i) {
      ss << "[";
      for (std::size_t j = 0; j < m_cols; ++j) ss << std::setfill(' ') << std::setw((int)width[j] + 1) << (*this)(i, j).toString() << " ";
      ss << "]";
      if (i + 1 < m_rows) ss << "\n";
    }
    return ss.str();
  }
  friend std::ostream& operator<<(std::ostream& os, const Matrix& m) { return os << m.toString(); }

  const std::vector<ELEMENT>& values() const { return m_values; }
  /// bytes of the elements' images, without any header (matrix.h:415-417)
  std::size_t byteSize() const { return m_rows * m_cols * ELEMENT::byteSize(); }

 private:
  Matrix(std::size_t r, std::size_t c, std::vector<ELEMENT> v) : m_rows(r), m_cols(c), m_values(std::move(v)) {}

  Matrix entrywise(const Matrix& o, Vector<ELEMENT> (Vector<ELEMENT>::*op)(const Vector<ELEMENT>&) const) const {
    if (rows() != o.rows() || cols() != o.cols()) throw std::invalid_argument("incompatible matrices");
    return Matrix(m_rows, m_cols, (Vector<ELEMENT>(m_values).*op)(Vector<ELEMENT>(o.m_values)).toStlVector());
  }

  std::size_t m_rows, m_cols;
  std::vector<ELEMENT> m_values;
};

// ---- linear systems (include/scl/math/matrix.h:585-828) -- what shamirRecoverC's Berlekamp-Welch is written against ----------
// Host scalars, any field (the batched Berlekamp-Welch solves its systems in LDS: kernels.hpp, k_bw_solve); the helpers
// keep the reference's behaviour on degenerate input -- which row counts as a pivot, what a free variable is set to.

/// the three elementary row operations (matrix.h:551-591)
template <typename ELEMENT>
void swapRows(Matrix<ELEMENT>& A, std::size_t k, std::size_t h) {
  if (k == h) return;
  for (std::size_t j = 0; j < A.cols(); ++j) std::swap(A(k, j), A(h, j));
}
template <typename ELEMENT>
void multiplyRow(Matrix<ELEMENT>& A, std::size_t row, const ELEMENT& m) {
  for (std::size_t j = 0; j < A.cols(); ++j) A(row, j) *= m;
}
/// row dst += m * row op
template <typename ELEMENT>
void addRows(Matrix<ELEMENT>& A, std::size_t dst, std::size_t op, const ELEMENT& m) {
  for (std::size_t j = 0; j < A.cols(); ++j) A(dst, j) += A(op, j) * m;
}

/// [A | B]: B's columns appended to A's (matrix.h:774-789)
template <typename ELEMENT>
Matrix<ELEMENT> createAugmentedMatrix(const Matrix<ELEMENT>& A, const Matrix<ELEMENT>& B) {
  const std::size_t n = A.rows(), m = A.cols(), k = B.cols();
  Matrix<ELEMENT> aug(n, m + k);
  for (std::size_t i = 0; i < n; ++i)
    for (std::size_t j = 0; j < m + k; ++j) aug(i, j) = j < m ? A(i, j) : B(i, j - m);
  return aug;
}
template <typename ELEMENT>
Matrix<ELEMENT> createAugmentedMatrix(const Matrix<ELEMENT>& A, const Vector<ELEMENT>& b) {
  return createAugmentedMatrix(A, b.toColumnMatrix());
}

/// reduced row echelon form in place (matrix.h:597-639): per column the first non-zero entry at or below the current row
/// becomes the pivot, its row is swapped up and scaled to a leading 1, every other row loses its multiple of it
template <typename ELEMENT>
void rowReduceInPlace(Matrix<ELEMENT>& A) {
  const std::size_t n = A.rows(), m = A.cols();
  const ELEMENT zero;
  for (std::size_t r = 0, c = 0; r < n && c < m; ++c) {
    std::size_t pivot = r;
    while (pivot < n && A(pivot, c) == zero) ++pivot;
    if (pivot == n) continue;  // nothing in this column: the row stays for the next one
    swapRows(A, pivot, r);
    multiplyRow(A, r, A(r, c).inverse());
    for (std::size_t k = 0; k < n; ++k) {
      if (k == r) continue;
      const ELEMENT t = A(k, c);
      if (t != zero) addRows(A, k, r, -t);
    }
    ++r;
  }
}

/// the row of the pivot of column `col` in a reduced matrix, -1 if the column has none (matrix.h:646-662): the LOWEST row
/// with a non-zero entry there, provided the columns 0 .. col - 2 of that row are zero (the reference's loop bound)
template <typename ELEMENT>
int getPivotInColumn(const Matrix<ELEMENT>& A, int col) {
  const ELEMENT zero = ELEMENT::zero();
  for (int i = static_cast<int>(A.rows()) - 1; i >= 0; --i) {
    if (A(i, col) == zero) continue;
    for (int k = 0; k < col - 1; ++k)
      if (A(i, k) != zero) return -1;
    return i;
  }
  return -1;
}

/// index of the last row that is not all zero (matrix.h:673-691); rows() - 1 .. 0, and size_t(-1) for a zero matrix
template <typename ELEMENT>
std::size_t findFirstNonZeroRow(const Matrix<ELEMENT>& A) {
  const ELEMENT zero = ELEMENT::zero();
  std::size_t row = A.rows();
  while (row-- > 0) {
    bool any = false;
    for (std::size_t j = 0; j < A.cols() && !any; ++j) any = A(row, j) != zero;
    if (any) break;
  }
  return row;
}

/// a solution read off a reduced augmented matrix (matrix.h:703-731): back-substitution from the last non-zero row; a free
/// variable is set to 1, the variables of dropped all-zero rows stay 0
template <typename ELEMENT>
Vector<ELEMENT> extractSolution(const Matrix<ELEMENT>& A) {
  const std::size_t n = A.rows(), m = A.cols();
  Vector<ELEMENT> x(m - 1);
  std::size_t i = findFirstNonZeroRow(A);
  for (int c = static_cast<int>(m) - 2 - static_cast<int>(n - i - 1); c >= 0; --c) {
    const int p = getPivotInColumn(A, c);
    if (p == -1) {
      x[static_cast<std::size_t>(c)] = ELEMENT{1};
      continue;
    }
    ELEMENT sum = ELEMENT::zero();
    for (std::size_t j = static_cast<std::size_t>(p) + 1; j < n; ++j) sum += A(i, j) * x[j];
    x[static_cast<std::size_t>(c)] = A(i, m - 1) - sum;
    --i;
  }
  return x;
}

/// does the reduced augmented matrix describe a solvable system (matrix.h:740-765)?  unique_only: no row of the coefficient
/// part may be all zero; otherwise only a zero row with a non-zero right-hand side rules a solution out
template <typename ELEMENT>
bool hasSolution(const Matrix<ELEMENT>& A, bool unique_only) {
  const ELEMENT zero;
  for (std::size_t i = 0; i < A.rows(); ++i) {
    bool all_zero = true;
    for (std::size_t j = 0; j + 1 < A.cols(); ++j) all_zero &= A(i, j) == zero;
    if (all_zero && (unique_only || A(i, A.cols() - 1) != zero)) return false;
  }
  return true;
}

/// A x = b with a unique solution: true and x set, false otherwise (matrix.h:811-828)
template <typename ELEMENT>
bool solveLinearSystem(Vector<ELEMENT>& x, const Matrix<ELEMENT>& A, const Vector<ELEMENT>& b) {
  if (A.rows() != b.size()) throw std::invalid_argument("malformed system of equations");
  auto aug = createAugmentedMatrix(A, b);
  rowReduceInPlace(aug);
  if (!hasSolution(aug, true)) return false;
  x = extractSolution(aug);
  return true;
}

}  // namespace scl::math

#endif
