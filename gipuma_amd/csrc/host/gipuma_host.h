// gipuma_host.h -- OpenCV-free host front-end in C++ (the reference's host language), on top of
// the C-ABI: everything the reference does between its command line and runcuda(), and between
// runcuda() and its result files.
//
//   command line                      main.cpp:164-428   -> parse_command_line
//   .P / Middlebury _par.txt readers  fileIoUtils.h:83-162
//   getCameraParameters               cameraGeometryUtils.h:174-353
//   selectViews                       main.cpp:430-499
//   depth range -> disparity range    main.cpp:898-906
//   disp.dmb / normals.dmb            fileIoUtils.h:320-368, main.cpp:1001-1015
//   result folder <out>/<timestamp>_<refname>/   main.cpp:717-723
//   3d_model0.ply                     displayUtils.h:78-159, main.cpp:1018-1025
//
// Images are read as binary PGM/PPM (8 bit) instead of through OpenCV's imread; PPM is converted
// with the BT.601 weights imread(IMREAD_GRAYSCALE) uses.  OpenCV's decomposeProjectionMatrix is
// restated (RQ with a positive diagonal, centre = -M^-1 p4) in double precision; neither is
// pinned by a reference test.
#pragma once
#include <string>
#include <vector>

#include "../../../include/gipuma_hip.h"

namespace gipuma_host {

struct AlgorithmParameters {  // reference algorithmparameters.h:21-51 (defaults) -- host view
    int algorithm = 0;
    float max_disparity = 256.0f, min_disparity = 0.0f;
    int box_hsize = 19, box_vsize = 19;
    float tau_color = 10.0f, tau_gradient = 2.0f, alpha = 0.9f, gamma = 10.0f;
    int border_value = -1, iterations = 8;
    bool color_processing = false;
    float dispTol = 1.0f, normTol = 0.1f, census_epsilon = 2.5f;
    int self_similarity_n = 50;
    float cam_scale = 1.0f;
    int num_img_processed = 1;
    float good_factor = 1.5f;
    int n_best = 2, cost_comb = GIPUMA_COMB_BEST_N;
    bool viewSelection = true;
    float depthMin = -1.0f, depthMax = -1.0f, min_angle = 5.0f, max_angle = 45.0f;
    float no_texture_sim = 0.9f, no_texture_per = 0.6f;
    unsigned max_views = 9;
    unsigned seed = 1;  // extension: --seed=
    unsigned mode_flags = 0;  // extension: --mode=exact|fast|literal -> GIPUMA_HIP_FLAG_FAST / _LITERAL (include/gipuma_hip.h)
};

struct InputFiles {  // reference main.h:39-54
    std::vector<std::string> img_filenames;
    std::string images_folder, p_folder, krt_file, calib_filename, camera_folder, bounding_folder, pmvs_folder,
        seed_file, gt_filename, gt_nocc_filename, occ_filename, gt_normal_filename;
};
struct GTcheckParameters {  // reference main.h:27-36
    bool gtCheck = false, noccCheck = false;
    float scale = 150.0f;
    float dispTolGT = 0.5f;
    float dispTolGT2 = 0.5f;  // (uninitialised in the reference unless --gtDepth_tolerance2= is given)
    float divFactor = 4.0f;   // ground-truth value / divFactor = disparity (Middlebury small: 4, KITTI: 255)
};
struct GtReport {  // what computeError / computeNormalError report (groundTruthUtils.h:22-135)
    int num_gt = 0;
    float error = 0, error2 = 0, error_nocc = 0, error_valid = 0, error_valid_all = 0, valid_ratio = 0;
    float normal_error = 0, normal_error2 = 0;
    bool has_normals = false;
};
struct OutputFiles {  // reference main.h:57-61
    std::string parentFolder = "results", disparity_filename;
};

// returns 0, or -1 like getParametersFromCommandLine
int parse_command_line(int argc, char **argv, InputFiles &in, OutputFiles &out, AlgorithmParameters &ap,
                       GTcheckParameters *gt = nullptr);
// ground-truth maps: .dmb, .pfm (PF/Pf, bottom-up) or 8/16-bit PGM taken unchanged (imread(-1)), main.cpp:760-776
bool read_gt_map(const std::string &path, std::vector<float> &img, int &rows, int &cols);
// 16-bit PPM of (n * 32767 + 32767), RGB: unit normals, zero where all components are 32767 (main.cpp:799-817)
bool read_gt_normals(const std::string &path, std::vector<float> &n3, int &rows, int &cols);
// computeError (groundTruthUtils.h:22-95).  gt_nocc: the map handed over as `occImg` (converted to 8 bit like
// the reference's implicit Mat_<uint8_t> conversion); valid: per-pixel flags or nullptr (the reference passes zeros)
void compute_error(const float *gt, const float *gt_nocc, const float *disp, const unsigned char *valid, int rows,
                   int cols, const GTcheckParameters &g, GtReport &r);
// computeNormalError (groundTruthUtils.h:97-135), tolerances 0.2 / 0.3 rad as at main.cpp:1110
void compute_normal_error(const float *normals3, const float *gt3, int rows, int cols, float tol, float tol2, GtReport &r);

bool read_p_file(const std::string &path, double P[12]);
// fills P for the images named in `names`; returns the number found
int read_middlebury_par(const std::string &path, const std::vector<std::string> &names, std::vector<double> &P);

struct CameraSet {
    std::vector<gipuma_hip_camera> cams;  // index 0 = reference
    std::vector<double> P;                // 12 per view: K0 [R|t] after re-centring
    std::vector<double> C;                // 3 per view: centre of P
    float f = 0.0f;
};
// transformP = false keeps the cameras in the world frame (getCameraParameters(..., false), used for the PLY)
void get_camera_parameters(const std::vector<double> &P_list, int n, float cam_scale, CameraSet &cs,
                           bool transformP = true);
// from_bundler_get_range (main.cpp:89-118); false if the file holds no readable 3d point
bool bundler_depth_range(const std::string &path, const CameraSet &cs, AlgorithmParameters &ap);
// returns the selected subset; fills depthMin/depthMax when they are -1
std::vector<int> select_views(const CameraSet &cs, int cols, int rows, AlgorithmParameters &ap);

bool read_pnm_gray(const std::string &path, std::vector<float> &img, int &rows, int &cols);
// -color_processing: P6 only, float4 texels B, G, R, 0 (imread(IMREAD_COLOR) order, main.cpp:943-956)
bool read_pnm_colour(const std::string &path, std::vector<float> &img, int &rows, int &cols);
// imread(IMREAD_GRAYSCALE) / imread(IMREAD_COLOR) of main.cpp:741-744 for binary PNM and PNG files (by content)
bool read_image_gray(const std::string &path, std::vector<float> &img, int &rows, int &cols);
bool read_image_colour(const std::string &path, std::vector<float> &img, int &rows, int &cols);
int write_dmb(const std::string &path, const float *data, int rows, int cols, int nb);
bool read_dmb(const std::string &path, std::vector<float> &data, int &rows, int &cols, int &nb);
// 3d_model<i>.ply of storePlyFileBinary (displayUtils.h:78-159): per pixel the world point of its depth,
// the normal and the gray value three times; x outer, y inner.  `cam` = the NOT re-centred camera.
int write_ply_binary(const std::string &path, const float *depth, const float *normals3, const float *gray,
                     int gray_stride, int rows, int cols, const gipuma_hip_camera &cam);

// the whole of runGipuma (main.cpp:694-1199) minus visualisation: returns 0 and the folder written
int run_gipuma(const InputFiles &in, const OutputFiles &out, AlgorithmParameters &ap, std::string *folder,
               const GTcheckParameters *gt = nullptr);

}  // namespace gipuma_host

extern "C" {
// test hooks (ctypes): same math as the CLI uses
int gipuma_host_camera_parameters(const double *P_list, int n, float cam_scale, gipuma_hip_camera *out, float *f);
int gipuma_host_select_views(const double *P_list, int n, float cam_scale, int cols, int rows, float min_angle,
                             float max_angle, unsigned max_views, float *depth_min, float *depth_max, int *subset);
int gipuma_host_write_dmb(const char *path, const float *data, int rows, int cols, int nb);
int gipuma_host_write_ply(const char *path, const float *depth, const float *normals3, const float *gray, int rows,
                          int cols, const gipuma_hip_camera *cam);
int gipuma_host_camera_parameters_world(const double *P_list, int n, float cam_scale, gipuma_hip_camera *out);
int gipuma_host_main(int argc, char **argv);
/* image files as the CLI reads them (PNG or binary PNM): colour == 0: rows*cols floats; else rows*cols*4 floats B, G, R, 0.
 * Call with out == NULL to get the size. Returns 0, or -1 if the file cannot be read. */
int gipuma_host_read_image(const char *path, int colour, float *out, int *rows, int *cols);
// computeError / computeNormalError on caller data (tests): out = error, error2, error_nocc, error_valid,
// error_valid_all, valid_ratio, num_gt;  out2 = normal error, normal error2
// from_bundler_get_range on caller data (tests): *depth_min / *depth_max in = current values (-1 = unset)
int gipuma_host_bundler_depth_range(const char *path, const double *P_list, int n, float cam_scale, float *depth_min,
                                    float *depth_max);
// the ground-truth map reader of the CLI (.dmb / .pfm / .pgm): rows * cols floats to `out` (may be NULL to query
// the size)
int gipuma_host_read_gt_map(const char *path, float *out, int *rows, int *cols);
int gipuma_host_compute_error(const float *gt, const float *gt_nocc, const float *disp, const unsigned char *valid,
                              int rows, int cols, float div_factor, float tol, float tol2, float *out);
int gipuma_host_compute_normal_error(const float *normals3, const float *gt3, int rows, int cols, float tol,
                                     float tol2, float *out2);
}
