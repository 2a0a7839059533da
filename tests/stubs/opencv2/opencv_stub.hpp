// opencv_stub.hpp -- TEST INFRASTRUCTURE, not product code.
//
// Declarations (no definitions) of the small part of OpenCV's C++ API that the reference's host code uses
// (main.cpp, cameraGeometryUtils.h, fileIoUtils.h, displayUtils.h, groundTruthUtils.h, mathUtils.h), so that
// tests/test_adapter.py can run the compiler's front end over the reference's UNMODIFIED main.cpp against
// gipuma_amd/csrc/adapter/cuda_compat/ (hipcc -fsyntax-only): the image has no OpenCV, and the claim to check is
// that main.cpp needs nothing from CUDA that the compat layer does not provide.  Nothing here is ever linked.
#pragma once
#include <cfloat>
#include <climits>
#include <cstddef>
#include <cstdint>
#include <iostream>
#include <string>
#include <vector>

#define CV_MAJOR_VERSION 3
#define CV_8U 0
#define CV_16U 2
#define CV_32F 5
#define CV_64F 6
#define CV_MAKETYPE(depth, cn) ((depth) + (((cn) - 1) << 3))
#define CV_8UC1 CV_MAKETYPE(CV_8U, 1)
#define CV_8UC3 CV_MAKETYPE(CV_8U, 3)
#define CV_16UC1 CV_MAKETYPE(CV_16U, 1)
#define CV_16UC3 CV_MAKETYPE(CV_16U, 3)
#define CV_32FC1 CV_MAKETYPE(CV_32F, 1)
#define CV_32FC3 CV_MAKETYPE(CV_32F, 3)
#define CV_32FC4 CV_MAKETYPE(CV_32F, 4)
#define CV_64FC1 CV_MAKETYPE(CV_64F, 1)

namespace cv {

typedef unsigned char uchar;
typedef unsigned short ushort;
typedef std::string String;

template <typename T> T saturate_cast(double v);

template <typename T, int N>
struct Vec {
    T val[N];
    Vec();
    Vec(T a);
    Vec(T a, T b);
    Vec(T a, T b, T c);
    Vec(T a, T b, T c, T d);
    T &operator[](int i);
    const T &operator[](int i) const;
    T &operator()(int i);
    const T &operator()(int i) const;
    T dot(const Vec &o) const;
    Vec cross(const Vec &o) const;
    Vec mul(const Vec &o) const;
    Vec &operator+=(const Vec &o);
    Vec &operator-=(const Vec &o);
    Vec &operator*=(double s);
    Vec &operator/=(double s);
    template <typename U> operator Vec<U, N>() const;
};
template <typename T, int N> Vec<T, N> operator+(const Vec<T, N> &a, const Vec<T, N> &b);
template <typename T, int N> Vec<T, N> operator-(const Vec<T, N> &a, const Vec<T, N> &b);
template <typename T, int N> Vec<T, N> operator-(const Vec<T, N> &a);
template <typename T, int N> Vec<T, N> operator*(const Vec<T, N> &a, double s);
template <typename T, int N> Vec<T, N> operator*(double s, const Vec<T, N> &a);
template <typename T, int N> Vec<T, N> operator/(const Vec<T, N> &a, double s);
template <typename T, int N> bool operator==(const Vec<T, N> &a, const Vec<T, N> &b);
template <typename T, int N> bool operator!=(const Vec<T, N> &a, const Vec<T, N> &b);
template <typename T, int N> std::ostream &operator<<(std::ostream &o, const Vec<T, N> &v);
typedef Vec<uchar, 3> Vec3b;
typedef Vec<uchar, 4> Vec4b;
typedef Vec<ushort, 3> Vec3w;
typedef Vec<int, 2> Vec2i;
typedef Vec<int, 3> Vec3i;
typedef Vec<float, 2> Vec2f;
typedef Vec<float, 3> Vec3f;
typedef Vec<float, 4> Vec4f;
typedef Vec<double, 2> Vec2d;
typedef Vec<double, 3> Vec3d;
typedef Vec<double, 4> Vec4d;

template <typename T> struct Point_ {
    T x, y;
    Point_();
    Point_(T x, T y);
};
typedef Point_<int> Point;
typedef Point_<int> Point2i;
typedef Point_<float> Point2f;
typedef Point_<double> Point2d;
template <typename T> struct Point3_ {
    T x, y, z;
    Point3_();
    Point3_(T x, T y, T z);
};
typedef Point3_<float> Point3f;
template <typename T> struct Size_ {
    T width, height;
    Size_();
    Size_(T w, T h);
};
typedef Size_<int> Size;
template <typename T> struct Rect_ {
    T x, y, width, height;
    Rect_();
    Rect_(T x, T y, T w, T h);
};
typedef Rect_<int> Rect;
struct Range {
    int start, end;
    Range();
    Range(int s, int e);
    static Range all();
};
template <typename T> struct Scalar_ {
    T val[4];
    Scalar_();
    Scalar_(T a);
    Scalar_(T a, T b, T c = 0, T d = 0);
    static Scalar_ all(T v);
    T &operator[](int i);
};
typedef Scalar_<double> Scalar;

struct MatExpr;
struct Mat;
template <typename T> struct Mat_;
// the proxies OpenCV passes arrays through: an output array also binds a temporary matrix header
struct _InputArray {
    _InputArray(const Mat &m);
    _InputArray(const MatExpr &e);
    template <typename T> _InputArray(const std::vector<T> &v);
    _InputArray(double v);
};
struct _OutputArray {
    _OutputArray(Mat &m);
    _OutputArray(const Mat &m);  // (a temporary header: M.col(i), M(Range, Range), M(Rect))
    template <typename T> _OutputArray(std::vector<T> &v);
};
typedef const _InputArray &InputArray;
typedef const _OutputArray &OutputArray;
typedef const _InputArray &InputArrayOfArrays;
typedef const _OutputArray &OutputArrayOfArrays;
struct MatStep {
    size_t operator[](int i) const;
    operator size_t() const;
};
struct MatSize {
    int operator[](int i) const;
    Size operator()() const;
};

struct Mat {
    int flags, dims, rows, cols;
    uchar *data;
    MatSize size;
    MatStep step;
    Mat();
    Mat(int rows, int cols, int type);
    Mat(int rows, int cols, int type, const Scalar &s);
    Mat(int rows, int cols, int type, void *data, size_t step = 0);
    Mat(Size size, int type);
    Mat(Size size, int type, const Scalar &s);
    Mat(const Mat &m);
    Mat(const Mat &m, const Rect &roi);
    Mat(const Mat &m, const Range &rowRange, const Range &colRange = Range::all());
    Mat(const MatExpr &e);
    template <typename T, int N> explicit Mat(const Vec<T, N> &v, bool copy = true);
    template <typename T> explicit Mat(const std::vector<T> &v, bool copy = false);
    ~Mat();
    Mat &operator=(const Mat &m);
    Mat &operator=(const MatExpr &e);
    Mat &operator=(const Scalar &s);
    static MatExpr zeros(int rows, int cols, int type);
    static MatExpr zeros(Size size, int type);
    static MatExpr ones(int rows, int cols, int type);
    static MatExpr ones(Size size, int type);
    static MatExpr eye(int rows, int cols, int type);
    void create(int rows, int cols, int type);
    void create(Size size, int type);
    void release();
    Mat clone() const;
    void copyTo(OutputArray m) const;
    void copyTo(OutputArray m, InputArray mask) const;
    void convertTo(OutputArray m, int rtype, double alpha = 1, double beta = 0) const;
    Mat &setTo(const Scalar &s);
    Mat &setTo(const Scalar &s, InputArray mask);
    Mat reshape(int cn, int rows = 0) const;
    MatExpr t() const;
    MatExpr inv(int method = 0) const;
    MatExpr mul(const Mat &m, double scale = 1) const;
    Mat cross(const Mat &m) const;
    double dot(const Mat &m) const;
    Mat row(int y) const;
    Mat col(int x) const;
    Mat rowRange(int a, int b) const;
    Mat rowRange(const Range &r) const;
    Mat colRange(int a, int b) const;
    Mat colRange(const Range &r) const;
    Mat operator()(const Rect &roi) const;
    Mat operator()(Range rowRange, Range colRange) const;
    bool empty() const;
    bool isContinuous() const;
    int type() const;
    int depth() const;
    int channels() const;
    size_t total() const;
    size_t elemSize() const;
    size_t step1(int i = 0) const;
    template <typename T> T &at(int i0);
    template <typename T> const T &at(int i0) const;
    template <typename T> T &at(int i0, int i1);
    template <typename T> const T &at(int i0, int i1) const;
    template <typename T> T &at(Point p);
    template <typename T> T *ptr(int i0 = 0);
    template <typename T> const T *ptr(int i0 = 0) const;
    uchar *ptr(int i0 = 0);
    void push_back(const Mat &m);
    template <typename T> operator Vec<T, 3>() const;
    template <typename T> operator Vec<T, 4>() const;
    template <typename T> operator Mat_<T>() const;
};

struct MatExpr {
    operator Mat() const;
    template <typename T> operator Mat_<T>() const;
    MatExpr t() const;
    MatExpr inv(int method = 0) const;
    MatExpr mul(const Mat &m, double scale = 1) const;
    Mat row(int y) const;
    Mat col(int x) const;
    template <typename T> T &at(int i0, int i1);
};
MatExpr operator+(const Mat &a, const Mat &b);
MatExpr operator+(const Mat &a, const Scalar &s);
MatExpr operator+(const Mat &a, const MatExpr &b);
MatExpr operator+(const MatExpr &a, const Mat &b);
MatExpr operator+(const MatExpr &a, const MatExpr &b);
MatExpr operator-(const Mat &a, const Mat &b);
MatExpr operator-(const Mat &a, const Scalar &s);
MatExpr operator-(const Scalar &s, const Mat &a);
MatExpr operator-(const Mat &a, const MatExpr &b);
MatExpr operator-(const MatExpr &a, const Mat &b);
MatExpr operator-(const MatExpr &a, const MatExpr &b);
MatExpr operator-(const Mat &a);
MatExpr operator-(const MatExpr &a);
MatExpr operator*(const Mat &a, const Mat &b);
MatExpr operator*(const Mat &a, double s);
MatExpr operator*(double s, const Mat &a);
MatExpr operator*(const MatExpr &a, const Mat &b);
MatExpr operator*(const Mat &a, const MatExpr &b);
MatExpr operator*(const MatExpr &a, const MatExpr &b);
MatExpr operator*(const MatExpr &a, double s);
MatExpr operator*(double s, const MatExpr &a);
MatExpr operator/(const Mat &a, double s);
MatExpr operator/(const Mat &a, const Mat &b);
MatExpr operator/(double s, const Mat &a);
MatExpr operator/(const MatExpr &a, double s);
MatExpr operator<(const Mat &a, double s);
MatExpr operator>(const Mat &a, double s);
MatExpr operator<=(const Mat &a, double s);
MatExpr operator>=(const Mat &a, double s);
MatExpr operator==(const Mat &a, double s);
MatExpr operator!=(const Mat &a, double s);
MatExpr operator==(const Mat &a, const Mat &b);
MatExpr operator!=(const Mat &a, const Mat &b);
MatExpr operator&(const Mat &a, const Mat &b);
MatExpr operator|(const Mat &a, const Mat &b);
MatExpr operator&(const MatExpr &a, const MatExpr &b);
Mat &operator+=(Mat &a, const Mat &b);
Mat &operator-=(Mat &a, const Mat &b);
Mat &operator*=(Mat &a, double s);
Mat &operator/=(Mat &a, double s);
template <typename T, int N> MatExpr operator*(const Mat &a, const Vec<T, N> &v);
template <typename T, int N> MatExpr operator*(const MatExpr &a, const Vec<T, N> &v);
std::ostream &operator<<(std::ostream &o, const Mat &m);
std::ostream &operator<<(std::ostream &o, const MatExpr &m);

template <typename T>
struct Mat_ : public Mat {
    Mat_();
    Mat_(int rows, int cols);
    Mat_(int rows, int cols, const T &v);
    Mat_(Size size);
    Mat_(const Mat &m);
    Mat_(const MatExpr &e);
    template <int N> Mat_(const Vec<T, N> &v, bool copy = true);
    Mat_ &operator=(const Mat &m);
    Mat_ &operator=(const MatExpr &e);
    Mat_ &operator=(const T &v);
    T &operator()(int r, int c);
    const T &operator()(int r, int c) const;
    T &operator()(int i);
    const T &operator()(int i) const;
    T &operator()(Point p);
    T *operator[](int r);
    const T *operator[](int r) const;
    Mat_ clone() const;
    Mat_ row(int y) const;
    Mat_ col(int x) const;
    Mat_ operator()(const Rect &roi) const;
    Mat_ operator()(const Range &r, const Range &c) const;
    MatExpr t() const;
    MatExpr inv(int method = 0) const;
    template <int N> operator Vec<T, N>() const;
    // comma initialiser: Mat_<float> m = (Mat_<float>(3, 1) << a, b, c);
    struct Init {
        Init &operator,(T v);
        operator Mat_<T>() const;
        operator Mat() const;
    };
    Init operator<<(T v);
};
typedef Mat_<uchar> Mat1b;
typedef Mat_<float> Mat1f;
typedef Mat_<double> Mat1d;


enum { NORM_INF = 1, NORM_L1 = 2, NORM_L2 = 4, NORM_MINMAX = 32 };
enum { DECOMP_LU = 0, DECOMP_SVD = 1 };
enum { IMREAD_UNCHANGED = -1, IMREAD_GRAYSCALE = 0, IMREAD_COLOR = 1, IMREAD_ANYDEPTH = 2, IMREAD_ANYCOLOR = 4 };
enum { COLOR_BGR2RGB = 4, COLOR_RGB2BGR = 4, COLOR_BGR2GRAY = 6, COLOR_RGB2GRAY = 7, COLOR_GRAY2BGR = 8, COLOR_GRAY2RGB = 8, COLOR_BGR2BGRA = 0, COLOR_BGRA2BGR = 1, COLOR_RGB2RGBA = 0 };
enum { COLORMAP_AUTUMN = 0, COLORMAP_JET = 2, COLORMAP_HOT = 11 };
enum { INTER_NEAREST = 0, INTER_LINEAR = 1, INTER_CUBIC = 2, INTER_AREA = 3 };
enum { WINDOW_NORMAL = 0, WINDOW_AUTOSIZE = 1 };

double norm(InputArray a, int normType = NORM_L2);
double norm(InputArray a, InputArray b, int normType = NORM_L2);
double norm(const MatExpr &e, int normType = NORM_L2);
template <typename T, int N> double norm(const Vec<T, N> &v);
template <typename T, int N> double norm(const Vec<T, N> &a, const Vec<T, N> &b);
template <typename T, int N> Vec<T, N> normalize(const Vec<T, N> &v);
void normalize(InputArray src, OutputArray dst, double alpha = 1, double beta = 0, int normType = NORM_L2, int dtype = -1,
               InputArray mask = _InputArray(0.0));
void minMaxLoc(InputArray src, double *minVal, double *maxVal = 0, Point *minLoc = 0, Point *maxLoc = 0,
               InputArray mask = _InputArray(0.0));
template <typename M> void split(const Mat &m, std::vector<M> &mv);
void split(const Mat &m, Mat *mv);
template <typename M> void merge(const std::vector<M> &mv, OutputArray dst);
void merge(const Mat *mv, size_t count, OutputArray dst);
void LUT(InputArray src, InputArray lut, OutputArray dst);
void bitwise_and(InputArray a, InputArray b, OutputArray dst, InputArray mask = _InputArray(0.0));
void bitwise_or(InputArray a, InputArray b, OutputArray dst, InputArray mask = _InputArray(0.0));
void bitwise_not(InputArray a, OutputArray dst, InputArray mask = _InputArray(0.0));
void hconcat(InputArray a, InputArray b, OutputArray dst);
void hconcat(const std::vector<Mat> &src, OutputArray dst);
void vconcat(InputArray a, InputArray b, OutputArray dst);
void transpose(InputArray src, OutputArray dst);
double determinant(InputArray m);
double invert(InputArray src, OutputArray dst, int flags = DECOMP_LU);
Scalar mean(InputArray src, InputArray mask = _InputArray(0.0));
Scalar sum(InputArray src);
int countNonZero(InputArray src);
MatExpr abs(const Mat &m);
MatExpr abs(const MatExpr &e);
void sqrt(InputArray src, OutputArray dst);
void pow(InputArray src, double p, OutputArray dst);
void absdiff(InputArray a, InputArray b, OutputArray dst);
void resize(InputArray src, OutputArray dst, Size dsize, double fx = 0, double fy = 0, int interpolation = INTER_LINEAR);
void cvtColor(InputArray src, OutputArray dst, int code, int dstCn = 0);
void applyColorMap(InputArray src, OutputArray dst, int colormap);
void GaussianBlur(InputArray src, OutputArray dst, Size ksize, double sigmaX, double sigmaY = 0, int borderType = 4);
void decomposeProjectionMatrix(InputArray projMatrix, OutputArray cameraMatrix, OutputArray rotMatrix, OutputArray transVect);
void decomposeProjectionMatrix(InputArray projMatrix, OutputArray cameraMatrix, OutputArray rotMatrix, OutputArray transVect,
                               OutputArray rotMatrixX, OutputArray rotMatrixY, OutputArray rotMatrixZ, OutputArray eulerAngles);
Mat imread(const String &filename, int flags = IMREAD_COLOR);
bool imwrite(const String &filename, InputArray img, const std::vector<int> &params = std::vector<int>());
void imshow(const String &winname, InputArray mat);
int waitKey(int delay = 0);
void namedWindow(const String &winname, int flags = WINDOW_AUTOSIZE);
void destroyWindow(const String &winname);
void destroyAllWindows();
int64_t getTickCount();
double getTickFrequency();

}  // namespace cv
