/*
 * cv_mini.hpp -- a FUNCTIONAL miniature of the part of OpenCV that the reference's camera front-end uses:
 * getCameraParameters (cameraGeometryUtils.h:174-353) and selectViews (main.cpp:430-499) with their helpers.
 * TEST INFRASTRUCTURE (oracle/ref_shim/hostref/build_hostref.sh): the image has no OpenCV, and without it the
 * reference's host math can only be restated, never RUN.  With this header the reference's own source lines are
 * compiled and executed; tests/test_host_frontend.py compares every Camera_cu field they produce with the
 * product's front-ends (gipuma_amd/cameras.py, gipuma_amd/csrc/host/gipuma_host.cpp).
 *
 * OpenCV is a third-party dependency of the reference, not in its tree and not pinned (CMakeLists.txt:13,
 * `find_package(OpenCV REQUIRED)`); what is restated here is its documented behaviour for CV_32F matrices, as in
 * the 2.4 / 3.x / 4.x sources (the functions below did not change their arithmetic across those releases):
 *   * Mat headers share data; operator()(Range, Range), col(), colRange() are views; copyTo writes through a view;
 *   * A*B (gemm, CV_32F): products accumulated in double, one rounding to float per element;
 *     A/s, s*A, -A: scaled conversion, (float)((double)a * alpha) with alpha = 1/s, s, -1 in double; A-B, A+B: float;
 *   * inv() (DECOMP_LU): 3x3 closed form (adjugate / det3) in double; otherwise LU with partial pivoting in float;
 *     inv(DECOMP_SVD): OpenCV runs a float Jacobi SVD -- here the exact (pseudo-)inverse in double, equal to it up
 *     to float rounding (the reference uses it for R^-1 and the 3x4 P's pseudo-inverse only);
 *   * determinant (3x3): det3 in double;
 *   * Vec: dot in float, norm accumulated in double, normalize(v) = v * (1/norm) in double;
 *   * decomposeProjectionMatrix (calib3d, cvDecomposeProjectionMatrix + cvRQDecomp3x3): works on double copies;
 *     RQ by three Givens rotations (x, y, z), then the sign fix "diagonal entries of K except the last positive";
 *     the 4-vector T is the right null vector of P (OpenCV: last row of V^T of an SVD; here: signed 3x3 minors,
 *     the same direction -- only T(0..2)/T(3) is used, cameraGeometryUtils.h:259).
 * Nothing under gipuma_amd/ includes or links this.
 */
#pragma once
#include <cfloat>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iostream>
#include <memory>
#include <sstream>
#include <string>
#include <vector>

#define CV_MAJOR_VERSION 3
#define CV_32F 5
#define CV_64F 6

namespace cv {

enum { DECOMP_LU = 0, DECOMP_SVD = 1 };
enum { NORM_L2 = 4 };

struct Range {
    int start, end;
    Range(int s, int e) : start(s), end(e) {}
};

template <typename T, int N>
struct Vec {
    T val[N];
    Vec() { for (int i = 0; i < N; i++) val[i] = T(0); }
    Vec(T a, T b) { static_assert(N == 2, "Vec2"); val[0] = a; val[1] = b; }
    Vec(T a, T b, T c) { static_assert(N == 3, "Vec3"); val[0] = a; val[1] = b; val[2] = c; }
    T &operator[](int i) { return val[i]; }
    const T &operator[](int i) const { return val[i]; }
    T &operator()(int i) { return val[i]; }
    const T &operator()(int i) const { return val[i]; }
    T dot(const Vec &o) const {  /* Matx::dot: accumulated in T */
        T s = T(0);
        for (int i = 0; i < N; i++) s += val[i] * o.val[i];
        return s;
    }
};
template <typename T, int N> Vec<T, N> operator-(const Vec<T, N> &a, const Vec<T, N> &b)
{
    Vec<T, N> r;
    for (int i = 0; i < N; i++) r.val[i] = a.val[i] - b.val[i];
    return r;
}
template <typename T, int N> Vec<T, N> operator*(const Vec<T, N> &a, double s)
{
    Vec<T, N> r;
    for (int i = 0; i < N; i++) r.val[i] = (T)((double)a.val[i] * s);  /* saturate_cast<T>(a*alpha) */
    return r;
}
template <typename T, int N> double norm(const Vec<T, N> &v)
{
    double s = 0;
    for (int i = 0; i < N; i++) s += (double)v.val[i] * (double)v.val[i];
    return std::sqrt(s);
}
template <typename T, int N> double norm(const Vec<T, N> &a, const Vec<T, N> &b)
{   /* cv::norm(src1, src2), NORM_L2 of the difference: differences in float, squares accumulated in double */
    double s = 0;
    for (int i = 0; i < N; i++) {
        const double d = (double)(T)(a.val[i] - b.val[i]);
        s += d * d;
    }
    return std::sqrt(s);
}
template <typename T, int N> Vec<T, N> normalize(const Vec<T, N> &v)
{
    const double nv = norm(v);
    return v * (nv ? 1. / nv : 0.);
}
typedef Vec<float, 2> Vec2f;
typedef Vec<float, 3> Vec3f;
typedef Vec<double, 3> Vec3d;
typedef Vec<int, 2> Vec2i;
typedef Vec<unsigned char, 3> Vec3b;

/* CV_32F matrices only.  Headers share the element buffer. */
class Mat {
public:
    int rows = 0, cols = 0;
    int step = 0;          /* elements between rows */
    size_t off = 0;
    std::shared_ptr<std::vector<float> > buf;

    Mat() {}
    Mat(int r, int c) : rows(r), cols(c), step(c), off(0), buf(std::make_shared<std::vector<float> >((size_t)r * c, 0.0f)) {}
    static Mat zeros(int r, int c, int) { return Mat(r, c); }
    static Mat ones(int r, int c, int)
    {
        Mat m(r, c);
        for (float &v : *m.buf) v = 1.0f;
        return m;
    }
    static Mat eye(int r, int c, int)
    {
        Mat m(r, c);
        for (int i = 0; i < r && i < c; i++) m.at(i, i) = 1.0f;
        return m;
    }
    bool empty() const { return !buf || rows == 0 || cols == 0; }
    void release() { buf.reset(); rows = cols = step = 0; off = 0; }
    float &at(int i, int j) const { return (*buf)[off + (size_t)i * step + j]; }
    Mat view(int r0, int r1, int c0, int c1) const
    {
        Mat m;
        m.rows = r1 - r0; m.cols = c1 - c0; m.step = step; m.off = off + (size_t)r0 * step + c0; m.buf = buf;
        return m;
    }
    Mat operator()(const Range &r, const Range &c) const { return view(r.start, r.end, c.start, c.end); }
    Mat col(int j) const { return view(0, rows, j, j + 1); }
    Mat row(int i) const { return view(i, i + 1, 0, cols); }
    Mat colRange(int a, int b) const { return view(0, rows, a, b); }
    Mat clone() const
    {
        Mat m(rows, cols);
        for (int i = 0; i < rows; i++)
            for (int j = 0; j < cols; j++) m.at(i, j) = at(i, j);
        return m;
    }
    /* copyTo(OutputArray): an unallocated / differently sized destination is (re)allocated -- impossible through a
     * temporary here, and the reference only copies into views of the right size */
    void copyTo(const Mat &dst) const
    {
        if (dst.rows != rows || dst.cols != cols) { fprintf(stderr, "cv_mini: copyTo size mismatch\n"); abort(); }
        for (int i = 0; i < rows; i++)
            for (int j = 0; j < cols; j++) dst.at(i, j) = at(i, j);
    }
    Mat t() const
    {
        Mat m(cols, rows);
        for (int i = 0; i < rows; i++)
            for (int j = 0; j < cols; j++) m.at(j, i) = at(i, j);
        return m;
    }
    Mat inv(int method = DECOMP_LU) const;
};

inline Mat scaled(const Mat &a, double alpha)
{
    Mat m(a.rows, a.cols);
    for (int i = 0; i < a.rows; i++)
        for (int j = 0; j < a.cols; j++) m.at(i, j) = (float)((double)a.at(i, j) * alpha);
    return m;
}
inline Mat operator*(const Mat &a, const Mat &b)
{   /* gemm, CV_32F: double accumulator, one rounding */
    if (a.cols != b.rows) { fprintf(stderr, "cv_mini: gemm size mismatch\n"); abort(); }
    Mat m(a.rows, b.cols);
    for (int i = 0; i < a.rows; i++)
        for (int j = 0; j < b.cols; j++) {
            double s = 0;
            for (int k = 0; k < a.cols; k++) s += (double)a.at(i, k) * (double)b.at(k, j);
            m.at(i, j) = (float)s;
        }
    return m;
}
inline Mat operator*(double s, const Mat &a) { return scaled(a, s); }
inline Mat operator*(const Mat &a, double s) { return scaled(a, s); }
inline Mat operator/(const Mat &a, double s) { return scaled(a, 1. / s); }
inline Mat operator-(const Mat &a) { return scaled(a, -1.); }
inline Mat operator-(const Mat &a, const Mat &b)
{
    if (a.rows != b.rows || a.cols != b.cols) { fprintf(stderr, "cv_mini: subtract size mismatch\n"); abort(); }
    Mat m(a.rows, a.cols);
    for (int i = 0; i < a.rows; i++)
        for (int j = 0; j < a.cols; j++) m.at(i, j) = a.at(i, j) - b.at(i, j);
    return m;
}
inline Mat operator+(const Mat &a, const Mat &b)
{
    if (a.rows != b.rows || a.cols != b.cols) { fprintf(stderr, "cv_mini: add size mismatch\n"); abort(); }
    Mat m(a.rows, a.cols);
    for (int i = 0; i < a.rows; i++)
        for (int j = 0; j < a.cols; j++) m.at(i, j) = a.at(i, j) + b.at(i, j);
    return m;
}

inline double det3d(const double *m)  /* row-major 3x3 */
{
    return m[0] * (m[4] * m[8] - m[5] * m[7]) - m[1] * (m[3] * m[8] - m[5] * m[6]) + m[2] * (m[3] * m[7] - m[4] * m[6]);
}
inline double determinant(const Mat &m)
{
    if (m.rows != 3 || m.cols != 3) { fprintf(stderr, "cv_mini: determinant of a non-3x3 matrix\n"); abort(); }
    double d[9];
    for (int i = 0; i < 9; i++) d[i] = (double)m.at(i / 3, i % 3);
    return det3d(d);
}

inline Mat Mat::inv(int method) const
{
    const int n = rows;
    if (method == DECOMP_SVD) {
        /* (pseudo-)inverse in double: A^T (A A^T)^-1 for a full-row-rank rows x cols matrix, rows <= 3 */
        if (rows > 3 || cols < rows) { fprintf(stderr, "cv_mini: inv(DECOMP_SVD) shape\n"); abort(); }
        double g[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
        for (int i = 0; i < rows; i++)
            for (int j = 0; j < rows; j++) {
                double s = 0;
                for (int k = 0; k < cols; k++) s += (double)at(i, k) * (double)at(j, k);
                g[3 * i + j] = s;
            }
        const double d = det3d(g);
        double gi[9];
        gi[0] = (g[4] * g[8] - g[5] * g[7]) / d; gi[1] = (g[2] * g[7] - g[1] * g[8]) / d; gi[2] = (g[1] * g[5] - g[2] * g[4]) / d;
        gi[3] = (g[5] * g[6] - g[3] * g[8]) / d; gi[4] = (g[0] * g[8] - g[2] * g[6]) / d; gi[5] = (g[2] * g[3] - g[0] * g[5]) / d;
        gi[6] = (g[3] * g[7] - g[4] * g[6]) / d; gi[7] = (g[1] * g[6] - g[0] * g[7]) / d; gi[8] = (g[0] * g[4] - g[1] * g[3]) / d;
        Mat m(cols, rows);
        for (int i = 0; i < cols; i++)
            for (int j = 0; j < rows; j++) {
                double s = 0;
                for (int k = 0; k < rows; k++) s += (double)at(k, i) * gi[3 * k + j];
                m.at(i, j) = (float)s;
            }
        return m;
    }
    if (rows != cols) { fprintf(stderr, "cv_mini: inv() of a non-square matrix\n"); abort(); }
    if (n == 3) {  /* cv::invert, n == 3, CV_32F: adjugate over det3, in double */
        double s[9];
        for (int i = 0; i < 9; i++) s[i] = (double)at(i / 3, i % 3);
        double d = det3d(s);
        Mat m(3, 3);
        if (d != 0.) {
            d = 1. / d;
            m.at(0, 0) = (float)((s[4] * s[8] - s[5] * s[7]) * d);
            m.at(0, 1) = (float)((s[2] * s[7] - s[1] * s[8]) * d);
            m.at(0, 2) = (float)((s[1] * s[5] - s[2] * s[4]) * d);
            m.at(1, 0) = (float)((s[5] * s[6] - s[3] * s[8]) * d);
            m.at(1, 1) = (float)((s[0] * s[8] - s[2] * s[6]) * d);
            m.at(1, 2) = (float)((s[2] * s[3] - s[0] * s[5]) * d);
            m.at(2, 0) = (float)((s[3] * s[7] - s[4] * s[6]) * d);
            m.at(2, 1) = (float)((s[1] * s[6] - s[0] * s[7]) * d);
            m.at(2, 2) = (float)((s[0] * s[4] - s[1] * s[3]) * d);
        }
        return m;
    }
    /* hal::LU32f against the identity: partial pivoting, float arithmetic */
    std::vector<float> A((size_t)n * n), B((size_t)n * n, 0.0f);
    for (int i = 0; i < n; i++) {
        for (int j = 0; j < n; j++) A[(size_t)i * n + j] = at(i, j);
        B[(size_t)i * n + i] = 1.0f;
    }
    for (int i = 0; i < n; i++) {
        int k = i;
        for (int j = i + 1; j < n; j++)
            if (std::abs(A[(size_t)j * n + i]) > std::abs(A[(size_t)k * n + i])) k = j;
        if (std::abs(A[(size_t)k * n + i]) < FLT_EPSILON) return Mat(n, n);
        if (k != i)
            for (int j = 0; j < n; j++) {
                std::swap(A[(size_t)i * n + j], A[(size_t)k * n + j]);
                std::swap(B[(size_t)i * n + j], B[(size_t)k * n + j]);
            }
        const float d = -1 / A[(size_t)i * n + i];
        for (int j = i + 1; j < n; j++) {
            const float alpha = A[(size_t)j * n + i] * d;
            for (int c = i + 1; c < n; c++) A[(size_t)j * n + c] += alpha * A[(size_t)i * n + c];
            for (int c = 0; c < n; c++) B[(size_t)j * n + c] += alpha * B[(size_t)i * n + c];
        }
    }
    for (int i = n - 1; i >= 0; i--)
        for (int j = 0; j < n; j++) {
            float s = B[(size_t)i * n + j];
            for (int k = i + 1; k < n; k++) s -= A[(size_t)i * n + k] * B[(size_t)k * n + j];
            B[(size_t)i * n + j] = s / A[(size_t)i * n + i];
        }
    Mat m(n, n);
    for (int i = 0; i < n; i++)
        for (int j = 0; j < n; j++) m.at(i, j) = B[(size_t)i * n + j];
    return m;
}

template <typename T>
class Mat_ : public Mat {
public:
    Mat_() {}
    Mat_(const Mat &m) : Mat(m) {}
    Mat_ &operator=(const Mat &m) { Mat::operator=(m); return *this; }
    Mat_ operator()(const Range &r, const Range &c) const { return Mat_(Mat::operator()(r, c)); }  /* a view */
    T &operator()(int i, int j) const { return reinterpret_cast<T &>(at(i, j)); }
    T &operator()(int i) const { return reinterpret_cast<T &>(cols == 1 ? at(i, 0) : at(0, i)); }
    Mat_ clone() const { return Mat_(Mat::clone()); }
};

/* ---- calib3d ---- */
inline void matmul3(const double *a, const double *b, double *c, bool ta = false, bool tb = false)
{
    double r[9];
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) {
            double s = 0;
            for (int k = 0; k < 3; k++) s += (ta ? a[3 * k + i] : a[3 * i + k]) * (tb ? b[3 * j + k] : b[3 * k + j]);
            r[3 * i + j] = s;
        }
    for (int i = 0; i < 9; i++) c[i] = r[i];
}
inline void transpose3(double *m)
{
    std::swap(m[1], m[3]); std::swap(m[2], m[6]); std::swap(m[5], m[7]);
}
/* cvRQDecomp3x3 (calib3d/src/calibration.cpp): M = R * Q, R upper triangular, Q orthogonal */
inline void rq_decomp3x3(const double *Min, double *R, double *Q)
{
    double M[9], Qx[9], Qy[9], Qz[9];
    for (int i = 0; i < 9; i++) M[i] = Min[i];
    double s, c, z;
    /* Givens rotation for the x axis: zeroes m32 */
    s = M[7]; c = M[8];
    z = 1. / std::sqrt(c * c + s * s + DBL_EPSILON);
    c *= z; s *= z;
    { const double q[9] = {1, 0, 0, 0, c, s, 0, -s, c}; for (int i = 0; i < 9; i++) Qx[i] = q[i]; }
    matmul3(M, Qx, R);
    R[7] = 0;
    /* y axis: zeroes m31 */
    s = -R[6]; c = R[8];
    z = 1. / std::sqrt(c * c + s * s + DBL_EPSILON);
    c *= z; s *= z;
    { const double q[9] = {c, 0, -s, 0, 1, 0, s, 0, c}; for (int i = 0; i < 9; i++) Qy[i] = q[i]; }
    matmul3(R, Qy, M);
    M[6] = 0;
    /* z axis: zeroes m21 */
    s = M[3]; c = M[4];
    z = 1. / std::sqrt(c * c + s * s + DBL_EPSILON);
    c *= z; s *= z;
    { const double q[9] = {c, s, 0, -s, c, 0, 0, 0, 1}; for (int i = 0; i < 9; i++) Qz[i] = q[i]; }
    matmul3(M, Qz, R);
    R[3] = 0;
    /* the decomposition ambiguity: the diagonal entries of R, except the last one, shall be positive;
     * rotate by 180 degrees where necessary */
    if (R[0] < 0) {
        if (R[4] < 0) {  /* around z */
            R[0] *= -1; R[1] *= -1; R[4] *= -1;
            Qz[0] *= -1; Qz[1] *= -1; Qz[3] *= -1; Qz[4] *= -1;
        } else {         /* around y */
            R[0] *= -1; R[2] *= -1; R[5] *= -1; R[8] *= -1;
            transpose3(Qz);
            Qy[0] *= -1; Qy[2] *= -1; Qy[6] *= -1; Qy[8] *= -1;
        }
    } else if (R[4] < 0) {  /* around x */
        R[1] *= -1; R[2] *= -1; R[4] *= -1; R[5] *= -1; R[8] *= -1;
        transpose3(Qz);
        transpose3(Qy);
        Qx[4] *= -1; Qx[5] *= -1; Qx[7] *= -1; Qx[8] *= -1;
    }
    /* Q = Qz^T * Qy^T * Qx^T */
    double T[9];
    matmul3(Qz, Qy, T, true, true);
    matmul3(T, Qx, Q, false, true);
}
/* cvDecomposeProjectionMatrix: P = K [R | -R C]; T = homogeneous camera centre (right null vector of P) */
inline void decomposeProjectionMatrix(const Mat &P, Mat &K, Mat &Rm, Mat &T)
{
    if (P.rows != 3 || P.cols != 4) { fprintf(stderr, "cv_mini: decomposeProjectionMatrix needs a 3x4 matrix\n"); abort(); }
    double p[12], M[9], R[9], Q[9];
    for (int i = 0; i < 12; i++) p[i] = (double)P.at(i / 4, i % 4);
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) M[3 * i + j] = p[4 * i + j];
    rq_decomp3x3(M, R, Q);
    auto minor3 = [&](int c0, int c1, int c2) {
        const double m[9] = {p[c0], p[c1], p[c2], p[4 + c0], p[4 + c1], p[4 + c2], p[8 + c0], p[8 + c1], p[8 + c2]};
        return det3d(m);
    };
    double t[4] = {minor3(1, 2, 3), -minor3(0, 2, 3), minor3(0, 1, 3), -minor3(0, 1, 2)};
    const double nt = std::sqrt(t[0] * t[0] + t[1] * t[1] + t[2] * t[2] + t[3] * t[3]);
    K = Mat(3, 3);
    Rm = Mat(3, 3);
    T = Mat(4, 1);
    for (int i = 0; i < 9; i++) {
        K.at(i / 3, i % 3) = (float)R[i];
        Rm.at(i / 3, i % 3) = (float)Q[i];
    }
    for (int i = 0; i < 4; i++) T.at(i, 0) = (float)(t[i] / nt);
}

}  // namespace cv
