// gipuma_host.cpp -- see gipuma_host.h
#include "gipuma_host.h"

#include <dirent.h>
#include <sys/stat.h>
#include <zlib.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <ctime>
#include <fstream>
#include <iostream>
#include <limits>
#include <sstream>

namespace gipuma_host {

// ------------------------------------------------------------------------------------------ CLI
static bool starts(const char *a, const char *opt) { return strncmp(a, opt, strlen(opt)) == 0; }

int parse_command_line(int argc, char **argv, InputFiles &in, OutputFiles &out, AlgorithmParameters &ap,
                       GTcheckParameters *gt)
{
    GTcheckParameters gt_local;
    GTcheckParameters &g = gt ? *gt : gt_local;
    int camera_idx = 0;
    // options that take their value from the next argument (main.cpp:364-417): a trailing one is an error
    auto next = [&](int &i) -> const char * {
        if (i + 1 >= argc) {
            printf("Command-line parameter error: option %s needs a value\n", argv[i]);
            return nullptr;
        }
        return argv[++i];
    };
#define NEXT_INTO(dst)                 \
    {                                  \
        const char *v_ = next(i);      \
        if (!v_) return -1;            \
        dst = v_;                      \
    }
    // main.cpp:164-428: positional = image names (first = reference); --x=value; -flag value
    for (int i = 1; i < argc; i++) {
        const char *a = argv[i];
        auto val = [&](const char *opt) { return a + strlen(opt); };
        if (a[0] != '-') {
            in.img_filenames.push_back(a);
        } else if (starts(a, "--algorithm=")) {
            const char *v = val("--algorithm=");  // parsed for compatibility; the device path has PM_COST only
            const char *names[] = {"pm", "ct", "adct?", "ct_ss", "pm_ss", "adct", "adct_ss", "sct"};
            ap.algorithm = -1;
            for (int k = 0; k < 8; k++)
                if (!strcmp(v, names[k])) ap.algorithm = k;
            if (ap.algorithm < 0) {
                printf("Command-line parameter error: Unknown stereo algorithm\n\n");
                return -1;
            }
        } else if (starts(a, "--cost_comb=")) {
            const char *v = val("--cost_comb=");
            ap.cost_comb = !strcmp(v, "all") ? GIPUMA_COMB_ALL : !strcmp(v, "best_n") ? GIPUMA_COMB_BEST_N
                         : !strcmp(v, "angle") ? GIPUMA_COMB_ANGLE : !strcmp(v, "good") ? GIPUMA_COMB_GOOD : -1;
            if (ap.cost_comb < 0) {
                printf("Command-line parameter error: Unknown cost combination method\n\n");
                return -1;
            }
        } else if (starts(a, "--max-disparity=")) {
            if (sscanf(val("--max-disparity="), "%f", &ap.max_disparity) != 1 || ap.max_disparity < 1) {
                printf("Command-line parameter error: The max disparity (--maxdisparity=<...>) must be a positive integer \n");
                return -1;
            }
        } else if (starts(a, "--blocksize=")) {
            int k;
            if (sscanf(val("--blocksize="), "%d", &k) != 1 || k < 1 || k % 2 != 1) {
                printf("Command-line parameter error: The block size (--blocksize=<...>) must be a positive odd number\n");
                return -1;
            }
            ap.box_hsize = ap.box_vsize = k;
        }
#define FOPT(name, field) else if (starts(a, name)) sscanf(val(name), "%f", &ap.field);
#define IOPT(name, field) else if (starts(a, name)) sscanf(val(name), "%d", &ap.field);
        FOPT("--good_factor=", good_factor) FOPT("--cost_tau_color=", tau_color)
        FOPT("--cost_tau_gradient=", tau_gradient) FOPT("--cost_alpha=", alpha) FOPT("--cost_gamma=", gamma)
        IOPT("--border_value=", border_value) IOPT("--iterations=", iterations) FOPT("--disp_tol=", dispTol)
        FOPT("--norm_tol=", normTol) IOPT("--ss_n=", self_similarity_n) FOPT("--ct_eps=", census_epsilon)
        FOPT("--cam_scale=", cam_scale) IOPT("--num_img_processed=", num_img_processed) IOPT("--n_best=", n_best)
        FOPT("--depth_min=", depthMin) FOPT("--depth_max=", depthMax) FOPT("--min_angle=", min_angle)
        FOPT("--max_angle=", max_angle) FOPT("--no_texture_sim", no_texture_sim)
        FOPT("--no_texture_per", no_texture_per)
#undef FOPT
#undef IOPT
        else if (starts(a, "--max_views=")) sscanf(val("--max_views="), "%u", &ap.max_views);
        else if (starts(a, "--seed=")) sscanf(val("--seed="), "%u", &ap.seed);  // extension
        else if (starts(a, "--mode=")) {  // extension: the library's mode flags (default exact; the reference's own build
                                          // choice is --use_fast_math, CMakeLists.txt:23)
            const std::string m = val("--mode=");
            if (m == "exact") ap.mode_flags = 0;
            else if (m == "fast") ap.mode_flags = GIPUMA_HIP_FLAG_FAST;
            else if (m == "literal") ap.mode_flags = GIPUMA_HIP_FLAG_LITERAL;
            else { printf("--mode= takes exact, fast or literal\n"); return -1; }
        }
        else if (starts(a, "--gtDepth_divisionFactor=")) sscanf(val("--gtDepth_divisionFactor="), "%f", &g.divFactor);
        else if (starts(a, "--gtDepth_tolerance2=")) sscanf(val("--gtDepth_tolerance2="), "%f", &g.dispTolGT2);
        else if (starts(a, "--gtDepth_tolerance=")) sscanf(val("--gtDepth_tolerance="), "%f", &g.dispTolGT);
        else if (starts(a, "--camera_idx=")) sscanf(val("--camera_idx="), "%d", &camera_idx);  // pmvs mode only
        else if (starts(a, "--pmvs_folder")) NEXT_INTO(in.pmvs_folder)
        else if (!strcmp(a, "-view_selection")) ap.viewSelection = true;
        else if (!strcmp(a, "-color_processing")) ap.color_processing = true;
        else if (!strcmp(a, "-o")) NEXT_INTO(out.disparity_filename)
        else if (!strcmp(a, "-output_folder")) NEXT_INTO(out.parentFolder)
        else if (!strcmp(a, "-calib_file")) NEXT_INTO(in.calib_filename)
        else if (!strcmp(a, "-gt")) NEXT_INTO(in.gt_filename)
        else if (!strcmp(a, "-gt_nocc")) NEXT_INTO(in.gt_nocc_filename)
        else if (!strcmp(a, "-occl_mask")) NEXT_INTO(in.occ_filename)
        else if (!strcmp(a, "-gt_normal")) NEXT_INTO(in.gt_normal_filename)
        else if (!strcmp(a, "-images_folder")) NEXT_INTO(in.images_folder)
        else if (!strcmp(a, "-p_folder")) NEXT_INTO(in.p_folder)
        else if (!strcmp(a, "-krt_file")) NEXT_INTO(in.krt_file)
        else if (!strcmp(a, "-camera_folder")) NEXT_INTO(in.camera_folder)
        else if (!strcmp(a, "--initial_seed")) NEXT_INTO(in.seed_file)
        else if (!strcmp(a, "-bounding_folder")) NEXT_INTO(in.bounding_folder)
        else printf("Command-line parameter warning: unknown option %s\n", a);  // e.g. -no_display
    }
#undef NEXT_INTO
    // --pmvs_folder: images under <pmvs>/visualize/, projection matrices under <pmvs>/txt/ (main.cpp:409-417);
    // the depth range, unless given, comes from <pmvs>/bundle.rd.out (bundler_depth_range below)
    if (!in.pmvs_folder.empty()) {
        std::cout << "Using pmvs information inside directory " << in.pmvs_folder << std::endl;
        in.images_folder = in.pmvs_folder + "/visualize/";
        in.p_folder = in.pmvs_folder + "/txt/";
        in.img_filenames.clear();
        int n_skipped = 0;
        if (DIR *dir = opendir(in.images_folder.c_str())) {
            while (dirent *ent = readdir(dir)) {
                if (!strcmp(ent->d_name, ".") || !strcmp(ent->d_name, "..")) continue;
                // (the reference takes every directory entry, main.cpp:131-141; entries that are not images
                //  in a format this front-end reads could only fail later, so they are left out here)
                const std::string nm(ent->d_name);
                const size_t dot = nm.find_last_of('.');
                std::string ext = dot == std::string::npos ? "" : nm.substr(dot + 1);
                for (char &ch : ext) ch = (char)tolower((unsigned char)ch);
                if (ext == "pgm" || ext == "ppm" || ext == "pnm" || ext == "pfm" || ext == "png")
                    in.img_filenames.push_back(nm);
                else
                    n_skipped++;
            }
            closedir(dir);
        } else {
            printf("Cannot open the image folder %s\n", in.images_folder.c_str());
        }
        if (n_skipped > 0)
            printf("%d file(s) in %s are in formats this front-end does not read (it reads png/pgm/ppm/pnm/pfm; a PMVS "
                   "visualize/ folder holds jpg): convert them, or use python -m gipuma_amd.batch, which reads jpg\n",
                   n_skipped, in.images_folder.c_str());
        if (in.img_filenames.empty()) {
            printf("No readable images in %s\n", in.images_folder.c_str());
            return -1;
        }
        std::sort(in.img_filenames.begin(), in.img_filenames.end());  // sorted like the reference's list (main.cpp:153)
        if (camera_idx < 0 || camera_idx >= (int)in.img_filenames.size()) {
            printf("Command-line parameter error: --camera_idx out of range\n");
            return -1;
        }
        std::cout << "Using image " << in.img_filenames[camera_idx] << " as reference camera" << std::endl;
        std::swap(in.img_filenames[0], in.img_filenames[camera_idx]);
    }
    std::cout << "Input files are: ";
    for (const auto &s : in.img_filenames) std::cout << s << " ";
    std::cout << std::endl;
    return 0;
}

// ------------------------------------------------------------------------------------------ readers
bool read_p_file(const std::string &path, double P[12])
{  // readPFileStrechaPmvs, fileIoUtils.h:83-110
    std::ifstream f(path.c_str());
    if (!f) return false;
    std::string line;
    int r = 0;
    while (r < 3 && std::getline(f, line)) {
        if (line.find("CONTOUR") != std::string::npos) continue;
        std::stringstream ss(line);
        double v[4];
        if (ss >> v[0] >> v[1] >> v[2] >> v[3]) {
            for (int c = 0; c < 4; c++) P[4 * r + c] = (double)(float)v[c];  // atof -> float
            r++;
        }
    }
    return r == 3;
}

int read_middlebury_par(const std::string &path, const std::vector<std::string> &names, std::vector<double> &P)
{  // readKRtFileMiddlebury, fileIoUtils.h:111-162: name K(9) R(9) t(3); P = K [R|t]
    std::ifstream f(path.c_str());
    if (!f) return 0;
    std::string line;
    std::getline(f, line);  // first line = count
    P.assign(names.size() * 12, 0.0);
    int found = 0;
    while (std::getline(f, line)) {
        std::stringstream ss(line);
        std::string name;
        double K[9], R[9], t[3];
        ss >> name;
        for (double &v : K) ss >> v;
        for (double &v : R) ss >> v;
        for (double &v : t) ss >> v;
        if (!ss) continue;
        for (size_t j = 0; j < names.size(); j++)
            if (names[j] == name) {
                double Rt[12];
                for (int r = 0; r < 3; r++) {
                    for (int c = 0; c < 3; c++) Rt[4 * r + c] = R[3 * r + c];
                    Rt[4 * r + 3] = t[r];
                }
                for (int r = 0; r < 3; r++)
                    for (int c = 0; c < 4; c++)
                        P[12 * j + 4 * r + c] = K[3 * r] * Rt[c] + K[3 * r + 1] * Rt[4 + c] + K[3 * r + 2] * Rt[8 + c];
                found++;
            }
    }
    return found;
}

// ------------------------------------------------------------------------------------------ 3x3 math
static void mul33(const double *a, const double *b, double *o)
{
    for (int r = 0; r < 3; r++)
        for (int c = 0; c < 3; c++) o[3 * r + c] = a[3 * r] * b[c] + a[3 * r + 1] * b[3 + c] + a[3 * r + 2] * b[6 + c];
}
static double det33(const double *m)
{
    return m[0] * (m[4] * m[8] - m[5] * m[7]) - m[1] * (m[3] * m[8] - m[5] * m[6]) + m[2] * (m[3] * m[7] - m[4] * m[6]);
}
static void inv33(const double *m, double *o)
{
    const double d = det33(m);
    o[0] = (m[4] * m[8] - m[5] * m[7]) / d; o[1] = (m[2] * m[7] - m[1] * m[8]) / d; o[2] = (m[1] * m[5] - m[2] * m[4]) / d;
    o[3] = (m[5] * m[6] - m[3] * m[8]) / d; o[4] = (m[0] * m[8] - m[2] * m[6]) / d; o[5] = (m[2] * m[3] - m[0] * m[5]) / d;
    o[6] = (m[3] * m[7] - m[4] * m[6]) / d; o[7] = (m[1] * m[6] - m[0] * m[7]) / d; o[8] = (m[0] * m[4] - m[1] * m[3]) / d;
}
static void mulv(const double *m, const double *v, double *o)
{
    for (int r = 0; r < 3; r++) o[r] = m[3 * r] * v[0] + m[3 * r + 1] * v[1] + m[3 * r + 2] * v[2];
}
static double dot(const double *a, const double *b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }
static double nrm(const double *a) { return std::sqrt(dot(a, a)); }

// decomposeProjectionMatrix (cameraGeometryUtils.h:252): M = K R by Gram-Schmidt from the last
// row up (the RQ decomposition with a positive diagonal), C = -M^-1 p4
static void decompose(const double *Pin, double *K, double *R, double *C)
{
    double P[12];
    memcpy(P, Pin, sizeof P);
    double M[9] = {P[0], P[1], P[2], P[4], P[5], P[6], P[8], P[9], P[10]};
    if (det33(M) < 0)  // a projection matrix is defined up to sign
        for (double &v : P) v = -v;
    const double m1[3] = {P[0], P[1], P[2]}, m2[3] = {P[4], P[5], P[6]}, m3[3] = {P[8], P[9], P[10]};
    double r1[3], r2[3], r3[3], t[3];
    memset(K, 0, 9 * sizeof(double));
    K[8] = nrm(m3);
    for (int k = 0; k < 3; k++) r3[k] = m3[k] / K[8];
    K[5] = dot(m2, r3);
    for (int k = 0; k < 3; k++) t[k] = m2[k] - K[5] * r3[k];
    K[4] = nrm(t);
    for (int k = 0; k < 3; k++) r2[k] = t[k] / K[4];
    K[2] = dot(m1, r3);
    K[1] = dot(m1, r2);
    for (int k = 0; k < 3; k++) t[k] = m1[k] - K[1] * r2[k] - K[2] * r3[k];
    K[0] = nrm(t);
    for (int k = 0; k < 3; k++) r1[k] = t[k] / K[0];
    for (int k = 0; k < 3; k++) { R[k] = r1[k]; R[3 + k] = r2[k]; R[6 + k] = r3[k]; }
    double Mi[9], Mm[9] = {P[0], P[1], P[2], P[4], P[5], P[6], P[8], P[9], P[10]}, p4[3] = {P[3], P[7], P[11]};
    inv33(Mm, Mi);
    mulv(Mi, p4, C);
    for (int k = 0; k < 3; k++) C[k] = -C[k];
}

static void put(float *dst, const double *src, int n)
{
    for (int k = 0; k < n; k++) dst[k] = (float)src[k];
}

void get_camera_parameters(const std::vector<double> &P_list, int n, float cam_scale, CameraSet &cs,
                           bool transformP)
{  // cameraGeometryUtils.h:174-353
    cs.cams.assign(n, gipuma_hip_camera());
    cs.P.assign(12 * n, 0.0);
    cs.C.assign(3 * n, 0.0);
    std::vector<double> K(9 * n), R(9 * n), C(3 * n), t(3 * n);
    for (int i = 0; i < n; i++) {
        decompose(&P_list[12 * i], &K[9 * i], &R[9 * i], &C[3 * i]);
        mulv(&R[9 * i], &C[3 * i], &t[3 * i]);                       // t = -R C, :264
        for (int k = 0; k < 3; k++) t[3 * i + k] = -t[3 * i + k];
    }
    auto scaleK = [&](const double *Kin, double *Ko) {                // scaleK, :136-147
        memcpy(Ko, Kin, 9 * sizeof(double));
        Ko[0] /= cam_scale; Ko[4] /= cam_scale; Ko[2] /= cam_scale; Ko[5] /= cam_scale;
    };
    double K0[9];
    scaleK(&K[0], K0);
    cs.f = (float)K0[0];
    // transform = [R0 t0; 0 1]^-1 = [R0^T  -R0^T t0]                 :109-115, :270-271
    double R0t[9], t0i[3];
    for (int r = 0; r < 3; r++)
        for (int c = 0; c < 3; c++) R0t[3 * r + c] = R[3 * c + r];
    mulv(R0t, &t[0], t0i);
    for (int k = 0; k < 3; k++) t0i[k] = -t0i[k];
    if (!transformP) {  // transform = identity: cameras stay in the world frame (:270)
        for (int k = 0; k < 9; k++) R0t[k] = (k % 4 == 0) ? 1.0 : 0.0;
        t0i[0] = t0i[1] = t0i[2] = 0.0;
    }
    for (int i = 0; i < n; i++) {
        double Ki[9], Kinv[9], Rn[9], tn[3], Pn[12], Minv[9], Mm[9], Rinv[9];
        scaleK(&K[9 * i], Ki);
        inv33(Ki, Kinv);
        mul33(&R[9 * i], R0t, Rn);                                    // transformCamera, :117-128
        mulv(&R[9 * i], t0i, tn);
        for (int k = 0; k < 3; k++) tn[k] += t[3 * i + k];
        for (int r = 0; r < 3; r++) {                                 // P' = K0 [R'|t']  (K of camera 0 for all)
            for (int c = 0; c < 3; c++)
                Pn[4 * r + c] = K0[3 * r] * Rn[c] + K0[3 * r + 1] * Rn[3 + c] + K0[3 * r + 2] * Rn[6 + c];
            Pn[4 * r + 3] = K0[3 * r] * tn[0] + K0[3 * r + 1] * tn[1] + K0[3 * r + 2] * tn[2];
        }
        for (int r = 0; r < 3; r++)
            for (int c = 0; c < 3; c++) Mm[3 * r + c] = Pn[4 * r + c];
        inv33(Mm, Minv);                                              // :301
        double p4[3] = {Pn[3], Pn[7], Pn[11]}, Cn[3];
        mulv(Minv, p4, Cn);                                           // centre of P', :130-133
        for (int k = 0; k < 3; k++) Cn[k] = -Cn[k];
        inv33(&R[9 * i], Rinv);                                       // :297
        memcpy(&cs.P[12 * i], Pn, sizeof Pn);
        memcpy(&cs.C[3 * i], Cn, sizeof Cn);
        gipuma_hip_camera &cam = cs.cams[i];
        put(cam.K, Ki, 9); put(cam.K_inv, Kinv, 9); put(cam.R, Rn, 9); put(cam.t, tn, 3);
        put(cam.M_inv, Minv, 9); put(cam.P_col34, p4, 3); put(cam.C, Cn, 3); put(cam.R_orig_inv, Rinv, 9);
        cam.fx = (float)K0[0]; cam.fy = (float)K0[4]; cam.f = (float)K0[0];
        cam.alpha = (float)K0[0] / (float)K0[4];
        cam.baseline = 0.54f;                                         // :305
        cam.depth_min = 2.0f; cam.depth_max = 20.0f;                  // camera.h:34,38
    }
}

static void view_vector(const CameraSet &cs, int i, int x, int y, double *v)
{  // getViewVector, cameraGeometryUtils.h:68-76
    const double *P = &cs.P[12 * i];
    double M[9] = {P[0], P[1], P[2], P[4], P[5], P[6], P[8], P[9], P[10]}, Mi[9];
    inv33(M, Mi);
    double p[3] = {x - P[3], y - P[7], 1.0 - P[11]}, X[3];
    mulv(Mi, p, X);
    for (int k = 0; k < 3; k++) v[k] = X[k] - cs.C[3 * i + k];
    const double n = nrm(v);
    for (int k = 0; k < 3; k++) v[k] /= n;
}

std::vector<int> select_views(const CameraSet &cs, int cols, int rows, AlgorithmParameters &ap)
{  // main.cpp:430-499
    const int n = (int)cs.cams.size();
    const int x = cols / 2, y = rows / 2;
    double vref[3];
    view_vector(cs, 0, x, y, vref);
    const double lo = ap.min_angle * M_PI / 180.0, hi = ap.max_angle * M_PI / 180.0;
    double dmin = 9999, dmax = 0;
    std::vector<int> subset;
    if (ap.viewSelection)
        printf("Accepting intersection angle of central rays from %f to %f degrees, use --min_angle=<angle> and --max_angle=<angle> to modify them\n",
               ap.min_angle, ap.max_angle);
    for (int i = 1; i < n; i++) {
        double v[3], d[3];
        view_vector(cs, i, x, y, v);
        for (int k = 0; k < 3; k++) d[k] = cs.C[k] - cs.C[3 * i + k];
        const double baseline = nrm(d);
        double c = dot(vref, v);
        c = c > 1 ? 1 : (c < -1 ? -1 : c);
        const double angle = std::acos(c);                            // getAngle, mathUtils.h:16-24
        if (angle > lo && angle < hi) {
            if (ap.viewSelection) subset.push_back(i);
            dmin = std::min(dmin, (baseline / 2.0) / std::sin(hi / 2.0));
            dmax = std::max(dmax, (baseline / 2.0) / std::sin(lo / 2.0));
        }
    }
    if (ap.depthMin == -1) ap.depthMin = (float)dmin;
    if (ap.depthMax == -1) ap.depthMax = (float)dmax;
    if (!ap.viewSelection) {
        subset.clear();
        for (int i = 1; i < n; i++) subset.push_back(i);
        return subset;
    }
    if (subset.size() >= ap.max_views) {
        // the reference shuffles with srand(time(0)) (main.cpp:491-496); here the order of the
        // command line decides, so that a run is reproducible
        printf("Too many camera, selecting only the first %u of them (modify with --max_views=<number>)\n", ap.max_views);
        subset.resize(ap.max_views);
    }
    return subset;
}

// parse_bundler_3d_points + from_bundler_get_range, main.cpp:45-118: the 3d points of a Bundler v0.3 file
// (https://www.cs.cornell.edu/~snavely/bundler/bundler-v0.4-manual.html#S6), the distance of each to the
// centre of every source camera, and depthMin = 0.6 * the smallest / depthMax = 1.2 * the largest -- each only
// if still -1.  As in the reference, this runs AFTER selectViews (main.cpp:870-875), which has already
// replaced every -1 (main.cpp:481-484): the file is read and its range never takes effect.  Kept literal --
// including the centres of the RE-CENTRED cameras measured against world-frame points.
bool bundler_depth_range(const std::string &path, const CameraSet &cs, AlgorithmParameters &ap)
{
    FILE *fp = fopen(path.c_str(), "r");
    if (!fp) return false;
    char line[512];
    unsigned num_cameras = 0, num_points = 0;
    int c = fgetc(fp);
    if (c == '#') {
        if (!fgets(line, sizeof line, fp)) { fclose(fp); return false; }
    } else if (c != EOF) {
        ungetc(c, fp);
    }
    if (!fgets(line, sizeof line, fp) || sscanf(line, "%u %u", &num_cameras, &num_points) != 2) { fclose(fp); return false; }
    for (unsigned i = 0; i < 5 * num_cameras; i++)  // <f k1 k2>, three rows of R, t
        if (!fgets(line, sizeof line, fp)) { fclose(fp); return false; }
    float min_depth = 9999, max_depth = 0;
    unsigned got = 0;
    for (unsigned i = 0; i < num_points; i++) {
        float X[3] = {0, 0, 0};
        if (!fgets(line, sizeof line, fp)) break;
        sscanf(line, "%f %f %f", &X[0], &X[1], &X[2]);
        if (!fgets(line, sizeof line, fp)) break;  // colour
        if (!fgets(line, sizeof line, fp)) break;  // view list
        got++;
        for (size_t v = 1; v < cs.cams.size(); v++) {
            const float dx = X[0] - (float)cs.C[3 * v], dy = X[1] - (float)cs.C[3 * v + 1], dz = X[2] - (float)cs.C[3 * v + 2];
            const float depth = std::sqrt(dx * dx + dy * dy + dz * dz);
            min_depth = std::min(depth, min_depth);
            max_depth = std::max(depth, max_depth);
        }
    }
    fclose(fp);
    if (ap.depthMin == -1) ap.depthMin = min_depth - min_depth * 0.4f;
    if (ap.depthMax == -1) ap.depthMax = max_depth + max_depth * 0.2f;
    return got > 0;
}

// ------------------------------------------------------------------------------------------ images, dmb
bool read_pnm_gray(const std::string &path, std::vector<float> &img, int &rows, int &cols)
{
    FILE *f = fopen(path.c_str(), "rb");
    if (!f) return false;
    char magic[3] = {0, 0, 0};
    int maxv = 0;
    auto next_int = [&](int &v) {
        int c = fgetc(f);
        while (c == '#' || isspace(c)) {
            if (c == '#') while (c != '\n' && c != EOF) c = fgetc(f);
            c = fgetc(f);
        }
        v = 0;
        while (isdigit(c)) { v = 10 * v + (c - '0'); c = fgetc(f); }
    };
    if (fread(magic, 1, 2, f) != 2 || magic[0] != 'P' || (magic[1] != '5' && magic[1] != '6')) { fclose(f); return false; }
    next_int(cols); next_int(rows); next_int(maxv);
    if (cols < 1 || rows < 1 || maxv != 255) { fclose(f); return false; }
    const int ch = magic[1] == '6' ? 3 : 1;
    std::vector<unsigned char> raw((size_t)rows * cols * ch);
    const bool ok = fread(raw.data(), 1, raw.size(), f) == raw.size();
    fclose(f);
    if (!ok) return false;
    img.resize((size_t)rows * cols);
    for (size_t k = 0; k < img.size(); k++) {
        if (ch == 1) img[k] = (float)raw[k];                          // main.cpp:941
        else {  // OpenCV's 8-bit BGR2GRAY: (R*4899 + G*9617 + B*1868 + 8192) >> 14
            const int r = raw[3 * k], g = raw[3 * k + 1], b = raw[3 * k + 2];
            img[k] = (float)((r * 4899 + g * 9617 + b * 1868 + 8192) >> 14);
        }
    }
    return true;
}

bool read_pnm_colour(const std::string &path, std::vector<float> &img, int &rows, int &cols)
{
    FILE *f = fopen(path.c_str(), "rb");
    if (!f) return false;
    int maxv = 0;
    auto next_int = [&](int &v) {
        int c = fgetc(f);
        while (c == '#' || isspace(c)) {
            if (c == '#') while (c != '\n' && c != EOF) c = fgetc(f);
            c = fgetc(f);
        }
        v = 0;
        while (isdigit(c)) { v = 10 * v + (c - '0'); c = fgetc(f); }
    };
    char magic[2];
    if (fread(magic, 1, 2, f) != 2 || magic[0] != 'P' || magic[1] != '6') { fclose(f); return false; }
    next_int(cols); next_int(rows); next_int(maxv);
    if (cols < 1 || rows < 1 || maxv != 255) { fclose(f); return false; }
    std::vector<unsigned char> raw((size_t)rows * cols * 3);
    const bool ok = fread(raw.data(), 1, raw.size(), f) == raw.size();
    fclose(f);
    if (!ok) return false;
    img.assign((size_t)rows * cols * 4, 0.0f);
    for (size_t k = 0; k < (size_t)rows * cols; k++) {
        img[4 * k + 0] = (float)raw[3 * k + 2];  // B
        img[4 * k + 1] = (float)raw[3 * k + 1];  // G
        img[4 * k + 2] = (float)raw[3 * k + 0];  // R
    }
    return true;
}

// ---- PNG (what the reference's scripts pass: scripts/dtu_fast.sh, templeRing.sh hand rect_*.png / *.png to imread,
// main.cpp:739-751).  The image has no libpng headers; the container format is simple enough to read with zlib alone:
// signature, IHDR / PLTE / IDAT / IEND chunks, one zlib stream, five scanline filters (PNG specification 1.2, section 6 and
// 9).  Bit depths 1-16, colour types gray / RGB / palette / gray+alpha / RGBA, non-interlaced.  16-bit samples keep their
// high byte and alpha is dropped, as imread without IMREAD_ANYDEPTH / IMREAD_UNCHANGED does.  out: rows*cols*3 bytes
// R, G, B (gray images replicated), `was_gray` = the file had no colour channels.  `gray16` (optional): for 16-bit
// colour files the gray value libpng forms BEFORE it chops to 8 bits -- its transform order for IMREAD_GRAYSCALE:
// rgb_to_gray on the 16-bit samples, (9797 r + 19234 g + 3737 b + 16384) >> 15, then the high byte -- one byte per pixel;
// left empty for every other file.
static bool read_png_rgb8(const std::string &path, std::vector<unsigned char> &rgb, int &rows, int &cols, bool &was_gray,
                          std::vector<unsigned char> *gray16 = nullptr)
{
    FILE *f = fopen(path.c_str(), "rb");
    if (!f) return false;
    std::vector<unsigned char> file;
    unsigned char buf[65536];
    size_t got;
    while ((got = fread(buf, 1, sizeof buf, f)) > 0) file.insert(file.end(), buf, buf + got);
    fclose(f);
    static const unsigned char sig[8] = {0x89, 'P', 'N', 'G', 0x0d, 0x0a, 0x1a, 0x0a};
    if (file.size() < 8 + 25 || memcmp(file.data(), sig, 8)) return false;
    auto be32 = [&](size_t o) -> uint32_t {
        return ((uint32_t)file[o] << 24) | ((uint32_t)file[o + 1] << 16) | ((uint32_t)file[o + 2] << 8) | file[o + 3];
    };
    uint32_t w = 0, h = 0;
    int depth = 0, ctype = -1, interlace = 0;
    std::vector<unsigned char> idat, plte;
    for (size_t o = 8; o + 12 <= file.size();) {
        const uint32_t len = be32(o);
        if (o + 12 + (size_t)len > file.size()) return false;
        const char *type = (const char *)&file[o + 4];
        const unsigned char *d = &file[o + 8];
        if (!memcmp(type, "IHDR", 4) && len >= 13) {
            w = be32(o + 8);
            h = be32(o + 12);
            depth = d[8];
            ctype = d[9];
            interlace = d[12];
        } else if (!memcmp(type, "PLTE", 4)) {
            plte.assign(d, d + len);
        } else if (!memcmp(type, "IDAT", 4)) {
            idat.insert(idat.end(), d, d + len);
        } else if (!memcmp(type, "IEND", 4)) {
            break;
        }
        o += 12 + (size_t)len;
    }
    const int ch = ctype == 0 ? 1 : ctype == 2 ? 3 : ctype == 3 ? 1 : ctype == 4 ? 2 : ctype == 6 ? 4 : 0;
    // (a header may claim any size: nothing is allocated for more than 2^28 pixels -- the library itself refuses frames of
    //  2^29 texels and more, include/gipuma_hip.h -- so a corrupt file is refused instead of exhausting memory)
    if (!ch || w < 1 || h < 1 || w > 65535 || h > 65535 || (uint64_t)w * h > (1ull << 28) || idat.empty()) return false;
    if (!(depth == 8 || depth == 16 || (depth < 8 && (ctype == 0 || ctype == 3) && (depth == 1 || depth == 2 || depth == 4))))
        return false;
    if (ctype == 3 && (depth > 8 || plte.size() < 3)) return false;
    if (interlace) {
        printf("%s: interlaced (Adam7) PNG is not read here; re-save it non-interlaced\n", path.c_str());
        return false;
    }
    const size_t bpp_bits = (size_t)ch * depth, stride = (w * bpp_bits + 7) / 8, bpp = bpp_bits >= 8 ? bpp_bits / 8 : 1;
    std::vector<unsigned char> raw((stride + 1) * (size_t)h);
    uLongf raw_len = (uLongf)raw.size();
    if (uncompress(raw.data(), &raw_len, idat.data(), (uLong)idat.size()) != Z_OK || raw_len != raw.size()) return false;
    // scanline filters: 0 none, 1 sub, 2 up, 3 average, 4 Paeth (PNG 1.2, 6.2 - 6.6)
    std::vector<unsigned char> zero(stride, 0);
    for (size_t y = 0; y < h; y++) {
        unsigned char *cur = &raw[y * (stride + 1) + 1];
        const unsigned char *up = y ? &raw[(y - 1) * (stride + 1) + 1] : zero.data();
        const int ft = raw[y * (stride + 1)];
        for (size_t x = 0; x < stride; x++) {
            const int a = x >= bpp ? cur[x - bpp] : 0, b = up[x], c = x >= bpp ? up[x - bpp] : 0;
            int pred = 0;
            if (ft == 1) pred = a;
            else if (ft == 2) pred = b;
            else if (ft == 3) pred = (a + b) >> 1;
            else if (ft == 4) {
                const int pp = a + b - c, pa = abs(pp - a), pb = abs(pp - b), pc = abs(pp - c);
                pred = (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c);
            } else if (ft != 0) return false;
            cur[x] = (unsigned char)(cur[x] + pred);
        }
    }
    rows = (int)h;
    cols = (int)w;
    was_gray = ctype == 0 || ctype == 4;
    rgb.resize((size_t)h * w * 3);
    const int maxv = (1 << (depth < 8 ? depth : 8)) - 1;
    const bool want16 = gray16 && depth == 16 && (ctype == 2 || ctype == 6);
    if (gray16) gray16->clear();
    if (want16) gray16->resize((size_t)h * w);
    for (size_t y = 0; y < h; y++) {
        const unsigned char *row = &raw[y * (stride + 1) + 1];
        for (size_t x = 0; x < w; x++) {
            if (want16) {
                auto s16 = [&](int k) -> uint32_t { return ((uint32_t)row[(x * ch + k) * 2] << 8) | row[(x * ch + k) * 2 + 1]; };
                const uint32_t r = s16(0), g = s16(1), b = s16(2);
                // (libpng leaves a pixel with r == g == b as it is)
                const uint32_t g16 = (r == g && g == b) ? r : (9797u * r + 19234u * g + 3737u * b + 16384u) >> 15;
                (*gray16)[y * w + x] = (unsigned char)(g16 >> 8);
            }
            auto sample = [&](int k) -> int {  // k-th sample of pixel x, reduced to 8 bits
                if (depth == 8) return row[x * ch + k];
                if (depth == 16) return row[(x * ch + k) * 2];  // high byte
                const size_t bit = x * depth;                    // sub-byte: one sample per pixel
                return (row[bit >> 3] >> (8 - depth - (bit & 7))) & maxv;
            };
            unsigned char *o = &rgb[(y * w + x) * 3];
            if (ctype == 3) {
                const size_t idx = (size_t)sample(0) * 3;
                if (idx + 3 > plte.size()) return false;
                o[0] = plte[idx]; o[1] = plte[idx + 1]; o[2] = plte[idx + 2];
            } else if (ch <= 2) {
                int v = sample(0);
                if (depth < 8) v = v * 255 / maxv;  // (libpng's expansion of 1/2/4-bit gray)
                o[0] = o[1] = o[2] = (unsigned char)v;
            } else {
                o[0] = (unsigned char)sample(0); o[1] = (unsigned char)sample(1); o[2] = (unsigned char)sample(2);
            }
        }
    }
    return true;
}

static bool is_png(const std::string &path)
{
    FILE *f = fopen(path.c_str(), "rb");
    if (!f) return false;
    unsigned char m[4] = {0, 0, 0, 0};
    const bool ok = fread(m, 1, 4, f) == 4 && m[0] == 0x89 && m[1] == 'P' && m[2] == 'N' && m[3] == 'G';
    fclose(f);
    return ok;
}

// imread(path, IMREAD_GRAYSCALE) / imread(path, IMREAD_COLOR) of main.cpp:741-744 for the containers this front-end reads:
// binary PNM and PNG, by content.  Gray from colour: PNM through OpenCV's own 8-bit BGR2GRAY (read_pnm_gray); PNG the way
// OpenCV's PNG decoder does it -- inside libpng, png_set_rgb_to_gray(1, 0.299, 0.587): 15-bit coefficients 9797 / 19234 /
// 3737, truncated (libpng 1.6 without gamma tables; OpenCV and libpng are dependencies the reference does not pin,
// CMakeLists.txt:13, so the last bit of a gray value converted from a colour file is theirs to choose).
bool read_image_gray(const std::string &path, std::vector<float> &img, int &rows, int &cols)
{
    if (!is_png(path)) return read_pnm_gray(path, img, rows, cols);
    std::vector<unsigned char> rgb, gray16;
    bool was_gray = false;
    if (!read_png_rgb8(path, rgb, rows, cols, was_gray, &gray16)) return false;
    img.resize((size_t)rows * cols);
    for (size_t k = 0; k < img.size(); k++) {
        const int r = rgb[3 * k], g = rgb[3 * k + 1], b = rgb[3 * k + 2];
        if (!gray16.empty())  // 16-bit colour: gray formed on the 16-bit samples, then the high byte (libpng's order)
            img[k] = (float)gray16[k];
        else
            img[k] = was_gray ? (float)r : (float)((9797 * r + 19234 * g + 3737 * b) >> 15);
    }
    return true;
}

bool read_image_colour(const std::string &path, std::vector<float> &img, int &rows, int &cols)
{
    if (!is_png(path)) return read_pnm_colour(path, img, rows, cols);
    std::vector<unsigned char> rgb;
    bool was_gray = false;
    if (!read_png_rgb8(path, rgb, rows, cols, was_gray)) return false;
    img.assign((size_t)rows * cols * 4, 0.0f);
    for (size_t k = 0; k < (size_t)rows * cols; k++) {
        img[4 * k + 0] = (float)rgb[3 * k + 2];  // B
        img[4 * k + 1] = (float)rgb[3 * k + 1];  // G
        img[4 * k + 2] = (float)rgb[3 * k + 0];  // R
    }
    return true;
}

int write_dmb(const std::string &path, const float *data, int rows, int cols, int nb)
{  // writeDmb / writeDmbNormal, fileIoUtils.h:320-368: int32 {type=1, h, w, nb} + h*w*nb float32
    FILE *f = fopen(path.c_str(), "wb");
    if (!f) { printf("Error opening file %s", path.c_str()); return -1; }
    const int32_t hdr[4] = {1, rows, cols, nb};
    fwrite(hdr, sizeof(int32_t), 4, f);
    fwrite(data, sizeof(float), (size_t)rows * cols * nb, f);
    fclose(f);
    return 0;
}

bool read_dmb(const std::string &path, std::vector<float> &data, int &rows, int &cols, int &nb)
{  // readDmb, fileIoUtils.h:247-319
    FILE *f = fopen(path.c_str(), "rb");
    if (!f) return false;
    int32_t hdr[4];
    bool ok = fread(hdr, sizeof(int32_t), 4, f) == 4 && hdr[0] == 1;
    if (ok) {
        rows = hdr[1]; cols = hdr[2]; nb = hdr[3];
        data.resize((size_t)rows * cols * nb);
        ok = fread(data.data(), sizeof(float), data.size(), f) == data.size();
    }
    fclose(f);
    return ok;
}

// ------------------------------------------------------------------------------------------ runGipuma
int write_ply_binary(const std::string &path, const float *depth, const float *normals3, const float *gray,
                     int gray_stride, int rows, int cols, const gipuma_hip_camera &cam)
{  // storePlyFileBinary, displayUtils.h:78-159 (the reference's OpenMP loop writes the vertices in
   // whatever order its threads reach the critical section; here: its loop order, x outer, y inner)
    FILE *f = fopen(path.c_str(), "wb");
    if (!f) return -1;
    fprintf(f, "ply\nformat binary_little_endian 1.0\nelement vertex %d\n", rows * cols);
    fprintf(f, "property float x\nproperty float y\nproperty float z\n");
    fprintf(f, "property float nx\nproperty float ny\nproperty float nz\n");
    fprintf(f, "property uchar red\nproperty uchar green\nproperty uchar blue\nend_header\n");
    std::vector<unsigned char> buf((size_t)rows * cols * 27);
    unsigned char *o = buf.data();
    for (int x = 0; x < cols; x++)
        for (int y = 0; y < rows; y++) {
            const size_t k = (size_t)y * cols + x;
            // get3Dpoint, cameraGeometryUtils.h:51-61: M_inv * (depth * (x, y, 1) - P.col(3))
            const float d = depth[k];
            const float v[3] = {d * (float)x - cam.P_col34[0], d * (float)y - cam.P_col34[1], d - cam.P_col34[2]};
            float X[3];
            for (int r = 0; r < 3; r++)
                X[r] = cam.M_inv[3 * r] * v[0] + cam.M_inv[3 * r + 1] * v[1] + cam.M_inv[3 * r + 2] * v[2];
            const float big = std::numeric_limits<float>::max();
            if (!(X[0] < big && X[0] > -big) || !(X[1] < big && X[1] > -big) || !(X[2] < big && X[2] >= -big))
                X[0] = X[1] = X[2] = 0.0f;
            memcpy(o, X, 12);
            memcpy(o + 12, normals3 + 3 * k, 12);
            const float g = gray[(size_t)y * gray_stride * cols + (size_t)x * gray_stride];
            o[24] = o[25] = o[26] = (unsigned char)g;  // the gray value three times (:141-143)
            o += 27;
        }
    const size_t w = fwrite(buf.data(), 1, buf.size(), f);
    fclose(f);
    return w == buf.size() ? 0 : -1;
}

// ------------------------------------------------------------------------------------------ ground truth
static bool read_pnm_raw(const std::string &path, std::vector<float> &img, int &rows, int &cols, int &chans)
{  // P5 / P6, 8 or 16 bit (big endian), values unchanged: what imread(path, -1) + convertTo(CV_32F) gives
    std::ifstream f(path.c_str(), std::ios::binary);
    if (!f) return false;
    std::string magic;
    f >> magic;
    if (magic != "P5" && magic != "P6") return false;
    chans = magic == "P6" ? 3 : 1;
    int vals[3], k = 0;
    while (k < 3) {
        f >> std::ws;
        if (f.peek() == '#') {
            std::string c;
            std::getline(f, c);
            continue;
        }
        if (!(f >> vals[k])) return false;
        k++;
    }
    f.get();
    cols = vals[0]; rows = vals[1];
    const int bytes = vals[2] > 255 ? 2 : 1;
    std::vector<unsigned char> raw((size_t)rows * cols * chans * bytes);
    f.read((char *)raw.data(), (std::streamsize)raw.size());
    if ((size_t)f.gcount() != raw.size()) return false;
    img.resize((size_t)rows * cols * chans);
    for (size_t i = 0; i < img.size(); i++)
        img[i] = bytes == 2 ? (float)((raw[2 * i] << 8) | raw[2 * i + 1]) : (float)raw[i];
    return true;
}

bool read_gt_map(const std::string &path, std::vector<float> &img, int &rows, int &cols)
{  // main.cpp:760-776
    const std::string ext = path.substr(path.find_last_of('.') + 1);
    if (ext == "dmb") {
        int nb;
        return read_dmb(path, img, rows, cols, nb) && nb == 1;
    }
    if (ext == "pfm") {  // readPfm, fileIoUtils.h:370-460: "Pf", width height, scale, rows bottom-up
        FILE *fp = fopen(path.c_str(), "rb");
        if (!fp) return false;
        char magic[8];
        double scale;
        if (fscanf(fp, "%7s %d %d %lf", magic, &cols, &rows, &scale) != 4 || magic[0] != 'P') {
            fclose(fp);
            return false;
        }
        fgetc(fp);
        const int ch = magic[1] == 'F' ? 3 : 1;
        img.assign((size_t)rows * cols, 0.0f);
        std::vector<float> line((size_t)cols * ch);
        for (int y = rows - 1; y >= 0; y--) {
            if (fread(line.data(), sizeof(float), line.size(), fp) != line.size()) {
                fclose(fp);
                return false;
            }
            // ('PF': the reference reads all three floats of a pixel into one variable and keeps the last,
            //  fileIoUtils.h:439-445)
            for (int x = 0; x < cols; x++) img[(size_t)y * cols + x] = line[(size_t)x * ch + (ch - 1)];
        }
        fclose(fp);
        return true;
    }
    int chans;
    return read_pnm_raw(path, img, rows, cols, chans) && chans == 1;
}

bool read_gt_normals(const std::string &path, std::vector<float> &n3, int &rows, int &cols)
{  // main.cpp:799-817: 16-bit RGB, component - 32767, normalised; all-zero vectors carry no ground truth
    int chans;
    std::vector<float> raw;
    if (!read_pnm_raw(path, raw, rows, cols, chans) || chans != 3) return false;
    n3.assign(raw.size(), 0.0f);
    for (size_t k = 0; k < raw.size(); k += 3) {
        const int a = (int)raw[k] - 32767, b = (int)raw[k + 1] - 32767, c = (int)raw[k + 2] - 32767;
        if (a == 0 && b == 0 && c == 0) continue;
        const float len = std::sqrt((float)a * a + (float)b * b + (float)c * c);
        n3[k] = a / len; n3[k + 1] = b / len; n3[k + 2] = c / len;
    }
    return true;
}

void compute_error(const float *gt, const float *gt_nocc, const float *disp, const unsigned char *valid, int rows,
                   int cols, const GTcheckParameters &g, GtReport &r)
{  // computeError, groundTruthUtils.h:22-95 (the error images it also paints are not produced here)
    float error = 0, error2 = 0, errorNocc = 0, errorValid = 0;
    int numGt = 0, numNocc = 0, numValid = 0;
    for (size_t k = 0; k < (size_t)rows * cols; k++) {
        const float d1 = gt[k] / g.divFactor;
        if (d1 == 0.0f || d1 == -1.0f) continue;  // no ground truth here
        numGt++;
        // `occImg` is a Mat_<uint8_t> built from the float map: saturate_cast<uchar>
        const float o = gt_nocc[k];
        const int o8 = o != o ? 0 : (int)std::min(255.0f, std::max(0.0f, std::nearbyint(o)));
        const bool nocc = o8 != 0;
        if (nocc) numNocc++;
        const bool validPixel = valid && valid[k] != 0;
        if (validPixel) numValid++;
        const float diff = std::fabs(d1 - disp[k]);
        if (diff >= g.dispTolGT) {
            error++;
            if (nocc) errorNocc++;
            if (validPixel) errorValid++;
        }
        if (diff >= g.dispTolGT2) error2++;
    }
    r.num_gt = numGt;
    r.error = error / (float)numGt;
    r.error2 = error2 / (float)numGt;
    r.error_nocc = errorNocc / (float)numNocc;
    r.error_valid_all = (errorValid + (float)(numGt - numValid)) / (float)numGt;
    r.error_valid = errorValid / (float)numValid;
    r.valid_ratio = (float)numValid / (float)numGt;
}

void compute_normal_error(const float *normals3, const float *gt3, int rows, int cols, float tol, float tol2, GtReport &r)
{  // computeNormalError, groundTruthUtils.h:97-135; getAngle, mathUtils.h:16-24
    int numGt = 0, e1 = 0, e2 = 0;
    for (size_t k = 0; k < (size_t)rows * cols; k++) {
        const float *gn = gt3 + 3 * k, *n = normals3 + 3 * k;
        if (gn[0] + gn[1] + gn[2] < 0.1f) continue;
        numGt++;
        float angle = std::acos(gn[0] * n[0] + gn[1] * n[1] + gn[2] * n[2]);
        if (angle != angle) angle = 0.0f;
        if (angle > tol) e1++;
        if (angle > tol2) e2++;
    }
    r.normal_error = (float)e1 / (float)numGt;
    r.normal_error2 = (float)e2 / (float)numGt;
    r.has_normals = true;
}

int run_gipuma(const InputFiles &in, const OutputFiles &out, AlgorithmParameters &ap, std::string *folder,
               const GTcheckParameters *gt_in)
{
    GTcheckParameters gtp = gt_in ? *gt_in : GTcheckParameters();
    if (in.img_filenames.size() < 2) {
        printf("Command-line parameter error: at least 2 images must be specified\n");
        return -1;
    }
    const int n = (int)in.img_filenames.size();
    // result folder <parent>/<YYYYMMDD_HHMMSS>_<refname>/, main.cpp:702-723
    time_t tt;
    time(&tt);
    tm *pt = localtime(&tt);
    mkdir(out.parentFolder.c_str(), 0777);
    const std::string &ref = in.img_filenames[0];
    const std::string ref_name = ref.size() > 4 ? ref.substr(0, ref.size() - 4) : ref;
    char outputFolder[512];
    snprintf(outputFolder, sizeof outputFolder, "%s/%04d%02d%02d_%02d%02d%02d_%s", out.parentFolder.c_str(),
             pt->tm_year + 1900, pt->tm_mon + 1, pt->tm_mday, pt->tm_hour, pt->tm_min, pt->tm_sec, ref_name.c_str());
    mkdir(outputFolder, 0777);

    std::vector<std::vector<float>> imgs(n);
    int rows = 0, cols = 0;
    for (int i = 0; i < n; i++) {
        int r, c;
        const std::string path = in.images_folder + in.img_filenames[i];
        const bool ok = ap.color_processing ? read_image_colour(path, imgs[i], r, c) : read_image_gray(path, imgs[i], r, c);
        if (!ok || (i && (r != rows || c != cols))) {
            printf("Image seems to be invalid\n");
            return -1;
        }
        rows = r; cols = c;
    }
    // ground truth (main.cpp:757-817)
    std::vector<float> gtDisp, gtDispNocc, gtNormals;
    if (!in.gt_filename.empty()) {
        gtp.gtCheck = true;
        printf("Opening GT image %s\n", in.gt_filename.c_str());
        int r, c;
        if (!read_gt_map(in.gt_filename, gtDisp, r, c) || r != rows || c != cols) {
            printf("cannot read ground truth %s (same size as the images, .dmb / .pfm / PGM)\n", in.gt_filename.c_str());
            return -1;
        }
    }
    if (!in.gt_nocc_filename.empty()) {
        if (!gtp.gtCheck) {
            printf("Command-line parameter error: Ground truth image (-gt) must be specified for use of nocc GT\n");
            return -1;
        }
        gtp.noccCheck = true;
        printf("Opening nocc GT image %s\n", in.gt_nocc_filename.c_str());
        int r, c;
        if (!read_gt_map(in.gt_nocc_filename, gtDispNocc, r, c) || r != rows || c != cols) return -1;
    } else if (!in.occ_filename.empty()) {
        if (!gtp.gtCheck) {
            printf("Command-line parameter error: Ground truth image (-gt) must be specified for use of occlusion mask\n");
            return -1;
        }
        printf("Opening Occlusion image %s\n", in.occ_filename.c_str());
        std::vector<float> occ;
        int r, c;
        if (!read_pnm_gray(in.occ_filename, occ, r, c) || r != rows || c != cols) return -1;
        gtDispNocc = gtDisp;  // getNoccGTimg: the ground truth where the mask is set, 0 elsewhere
        for (size_t k = 0; k < occ.size(); k++)
            if (occ[k] == 0.0f) gtDispNocc[k] = 0.0f;
    } else {
        gtDispNocc = gtDisp;
    }
    if (!in.gt_normal_filename.empty()) {
        int r, c;
        if (!read_gt_normals(in.gt_normal_filename, gtNormals, r, c) || r != rows || c != cols) {
            printf("cannot read ground-truth normals %s (16-bit PPM)\n", in.gt_normal_filename.c_str());
            return -1;
        }
    }

    std::vector<double> P(12 * n, 0.0);
    if (!in.krt_file.empty()) {
        if (read_middlebury_par(in.krt_file, in.img_filenames, P) != n) { printf("krt_file incomplete\n"); return -1; }
    } else if (!in.p_folder.empty()) {
        for (int i = 0; i < n; i++) {
            // <name>.P next to -p_folder, <name without extension>.txt in a pmvs tree (cameraGeometryUtils.h:199-222)
            std::string pf = in.p_folder + in.img_filenames[i] + ".P";
            if (!in.pmvs_folder.empty()) {
                const std::string &nm = in.img_filenames[i];
                pf = in.p_folder + nm.substr(0, nm.find_last_of('.')) + ".txt";
            }
            if (!read_p_file(pf, &P[12 * i])) {
                printf("cannot read %s\n", pf.c_str());
                return -1;
            }
        }
    } else {
        printf("need -p_folder or -krt_file\n");
        return -1;
    }
    CameraSet cs;
    get_camera_parameters(P, n, ap.cam_scale, cs);
    std::vector<int> subset = select_views(cs, cols, rows, ap);
    if (!in.pmvs_folder.empty()) {  // main.cpp:873-875
        const std::string bf = in.pmvs_folder + "/bundle.rd.out";
        std::cout << "Using bundler file " << bf << " to obtain depth range" << std::endl;
        if (!bundler_depth_range(bf, cs, ap)) printf("Warning: no 3d points read from %s\n", bf.c_str());
    }
    std::cout << "Total number of images used: " << subset.size() << std::endl;
    std::cout << "Selected views: ";
    for (int v : subset) std::cout << v << ", ";
    std::cout << std::endl;
    cs.cams[0].depth_min = ap.depthMin;                               // main.cpp:898-906
    cs.cams[0].depth_max = ap.depthMax;
    ap.min_disparity = cs.f * cs.cams[0].baseline / ap.depthMax;
    ap.max_disparity = cs.f * cs.cams[0].baseline / ap.depthMin;
    std::cout << "Range of Minimum/Maximum depth is: " << ap.min_disparity << " " << ap.max_disparity
              << ", change it with --depth_min=<value> and  --depth_max=<value>" << std::endl;

    std::vector<const float *> ptrs(n);
    for (int i = 0; i < n; i++) ptrs[i] = imgs[i].data();
    gipuma_hip_desc d{};
    d.abi_version = GIPUMA_HIP_ABI_VERSION;
    d.rows = rows; d.cols = cols; d.channels = ap.color_processing ? 4 : 1; d.pitch = cols * d.channels;
    d.n_images = n;
    d.images = ptrs.data(); d.cameras = cs.cams.data();
    d.n_selected = (int)subset.size(); d.selected = subset.data();
    d.params.box_hsize = ap.box_hsize; d.params.box_vsize = ap.box_vsize; d.params.iterations = ap.iterations;
    d.params.n_best = ap.n_best; d.params.cost_comb = ap.cost_comb; d.params.alpha = ap.alpha;
    d.params.tau_color = ap.tau_color; d.params.tau_gradient = ap.tau_gradient; d.params.gamma = ap.gamma;
    d.params.min_disparity = ap.min_disparity; d.params.max_disparity = ap.max_disparity;
    d.params.good_factor = ap.good_factor;
    d.seed = ap.seed;
    d.flags = ap.mode_flags;
    std::vector<float> norm4((size_t)rows * cols * 4), cost((size_t)rows * cols);
    gipuma_hip_timing t{};
    printf("Blocksize is %dx%d\n", ap.box_hsize, ap.box_vsize);
    printf("Number of iterations is %d\n", ap.iterations);
    const int rc = gipuma_hip_run(&d, norm4.data(), cost.data(), &t);
    if (rc) {
        fprintf(stderr, "gipuma_hip_run failed (%d): %s\n", rc, gipuma_hip_last_error());
        return rc;
    }
    printf("\t\tTotal time needed for computation: %f seconds\n", (t.ms_sweeps + t.ms_finalize) / 1000.f);
    // disp.dmb = norm4.w (depth), normals.dmb = world normals: main.cpp:976-985, 1001-1015
    std::vector<float> disp((size_t)rows * cols), nrm3((size_t)rows * cols * 3);
    for (size_t k = 0; k < disp.size(); k++) {
        disp[k] = norm4[4 * k + 3];
        nrm3[3 * k] = norm4[4 * k]; nrm3[3 * k + 1] = norm4[4 * k + 1]; nrm3[3 * k + 2] = norm4[4 * k + 2];
    }
    const std::string of(outputFolder);
    write_dmb(of + "/disp.dmb", disp.data(), rows, cols, 1);
    write_dmb(of + "/normals.dmb", nrm3.data(), rows, cols, 3);
    write_dmb(of + "/cost.dmb", cost.data(), rows, cols, 1);         // extra: the cost plane (cudacost.png in the reference)
    if (!out.disparity_filename.empty()) write_dmb(out.disparity_filename, disp.data(), rows, cols, 1);
    {   // 3d_model0.ply: points through the NOT re-centred camera 0, main.cpp:1018-1025
        CameraSet world;
        get_camera_parameters(P, n, ap.cam_scale, world, false);
        std::vector<float> gray;
        const float *g = imgs[0].data();
        int stride = 1;
        if (ap.color_processing) {  // the reference passes img_grayscale[] (BT.601 of B, G, R)
            gray.resize((size_t)rows * cols);
            for (size_t k = 0; k < gray.size(); k++)
                gray[k] = std::floor(0.114f * g[4 * k] + 0.587f * g[4 * k + 1] + 0.299f * g[4 * k + 2] + 0.5f);
            g = gray.data();
        }
        write_ply_binary(of + "/3d_model0.ply", disp.data(), nrm3.data(), g, stride, rows, cols, world.cams[0]);
    }
    if (gtp.gtCheck) {  // main.cpp:1086-1163
        GtReport rep;
        // the reference hands computeError an all-zero `valid` map (main.cpp:1089): no pixel passes the
        // occlusion check it stands for, so "valid" errors are 0/0 and "valid all" is 1
        std::vector<unsigned char> valid((size_t)rows * cols, 0);
        compute_error(gtDisp.data(), gtDispNocc.data(), disp.data(), valid.data(), rows, cols, gtp, rep);
        if (!gtNormals.empty()) compute_normal_error(nrm3.data(), gtNormals.data(), rows, cols, 0.2f, 0.3f, rep);
        std::ofstream rf((of + "/results.txt").c_str(), std::ios::out | std::ios::app);
        rf << "Valid pixels of GT subset: " << rep.valid_ratio << std::endl << std::endl;
        rf << "Error1: " << rep.error << std::endl << "Correct1: " << 1 - rep.error << std::endl;
        rf << "Error2: " << rep.error2 << std::endl << "Correct2: " << 1 - rep.error2 << std::endl;
        rf << "\n\nError (nocc): " << rep.error_nocc << std::endl;
        rf << "Error (valid occlusion check): " << rep.error_valid << std::endl;
        rf << "Error (valid occlusion check, div by #GT points): " << rep.error_valid_all << std::endl;
        // (the reference's median filter is commented out: its "after median filtering" figures repeat Error1/2)
        rf << "\nError after median filtering: " << rep.error << ", thresh 2: " << rep.error2 << std::endl;
        rf << "Correct %: " << 1 - rep.error << " (thresh 2: " << 1 - rep.error2 << ")" << std::endl;
        rf << "\nNormal error (0.2rad): " << rep.normal_error << std::endl;
        rf << "\nNormal error2 (0.3rad): " << rep.normal_error2 << std::endl;
        std::cout << "Error1: " << rep.error << std::endl << "Correct1: " << 1 - rep.error << std::endl;
        std::cout << "Error2: " << rep.error2 << std::endl << "Correct2: " << 1 - rep.error2 << std::endl;
        std::cout << "Error (nocc): " << rep.error_nocc << std::endl;
        std::cout << "Error (valid occlusion check): " << rep.error_valid << std::endl;
        std::cout << "Error (valid occlusion check, div by #GT points): " << rep.error_valid_all << std::endl;
    }
    if (folder) *folder = of;
    return 0;
}

}  // namespace gipuma_host

extern "C" {

int gipuma_host_camera_parameters(const double *P_list, int n, float cam_scale, gipuma_hip_camera *out, float *f)
{
    gipuma_host::CameraSet cs;
    gipuma_host::get_camera_parameters(std::vector<double>(P_list, P_list + 12 * n), n, cam_scale, cs);
    for (int i = 0; i < n; i++) out[i] = cs.cams[i];
    if (f) *f = cs.f;
    return 0;
}

int gipuma_host_select_views(const double *P_list, int n, float cam_scale, int cols, int rows, float min_angle,
                             float max_angle, unsigned max_views, float *depth_min, float *depth_max, int *subset)
{
    gipuma_host::CameraSet cs;
    gipuma_host::get_camera_parameters(std::vector<double>(P_list, P_list + 12 * n), n, cam_scale, cs);
    gipuma_host::AlgorithmParameters ap;
    ap.min_angle = min_angle; ap.max_angle = max_angle; ap.max_views = max_views;
    ap.depthMin = *depth_min; ap.depthMax = *depth_max;
    std::vector<int> s = gipuma_host::select_views(cs, cols, rows, ap);
    for (size_t k = 0; k < s.size(); k++) subset[k] = s[k];
    *depth_min = ap.depthMin; *depth_max = ap.depthMax;
    return (int)s.size();
}

int gipuma_host_bundler_depth_range(const char *path, const double *P_list, int n, float cam_scale, float *depth_min,
                                    float *depth_max)
{  // depth_min / depth_max: in = current values (-1 = unset), out = after from_bundler_get_range
    gipuma_host::CameraSet cs;
    gipuma_host::get_camera_parameters(std::vector<double>(P_list, P_list + 12 * (size_t)n), n, cam_scale, cs);
    gipuma_host::AlgorithmParameters ap;
    ap.depthMin = *depth_min;
    ap.depthMax = *depth_max;
    const bool ok = gipuma_host::bundler_depth_range(path, cs, ap);
    *depth_min = ap.depthMin;
    *depth_max = ap.depthMax;
    return ok ? 0 : -1;
}

int gipuma_host_read_gt_map(const char *path, float *out, int *rows, int *cols)
{
    std::vector<float> img;
    int r = 0, c = 0;
    if (!gipuma_host::read_gt_map(path, img, r, c)) return -1;
    *rows = r;
    *cols = c;
    if (out) memcpy(out, img.data(), img.size() * sizeof(float));
    return 0;
}

int gipuma_host_compute_error(const float *gt, const float *gt_nocc, const float *disp, const unsigned char *valid,
                              int rows, int cols, float div_factor, float tol, float tol2, float *out)
{
    gipuma_host::GTcheckParameters g;
    g.divFactor = div_factor; g.dispTolGT = tol; g.dispTolGT2 = tol2;
    gipuma_host::GtReport r;
    gipuma_host::compute_error(gt, gt_nocc ? gt_nocc : gt, disp, valid, rows, cols, g, r);
    out[0] = r.error; out[1] = r.error2; out[2] = r.error_nocc; out[3] = r.error_valid; out[4] = r.error_valid_all;
    out[5] = r.valid_ratio; out[6] = (float)r.num_gt;
    return 0;
}

int gipuma_host_compute_normal_error(const float *normals3, const float *gt3, int rows, int cols, float tol,
                                     float tol2, float *out2)
{
    gipuma_host::GtReport r;
    gipuma_host::compute_normal_error(normals3, gt3, rows, cols, tol, tol2, r);
    out2[0] = r.normal_error; out2[1] = r.normal_error2;
    return 0;
}

int gipuma_host_write_ply(const char *path, const float *depth, const float *normals3, const float *gray, int rows,
                          int cols, const gipuma_hip_camera *cam)
{
    return gipuma_host::write_ply_binary(path, depth, normals3, gray, 1, rows, cols, *cam);
}

int gipuma_host_camera_parameters_world(const double *P_list, int n, float cam_scale, gipuma_hip_camera *out)
{
    gipuma_host::CameraSet cs;
    gipuma_host::get_camera_parameters(std::vector<double>(P_list, P_list + 12 * n), n, cam_scale, cs, false);
    for (int i = 0; i < n; i++) out[i] = cs.cams[i];
    return 0;
}

int gipuma_host_write_dmb(const char *path, const float *data, int rows, int cols, int nb)
{
    return gipuma_host::write_dmb(path, data, rows, cols, nb);
}

int gipuma_host_read_image(const char *path, int colour, float *out, int *rows, int *cols)
{
    if (!path || !rows || !cols) return -1;
    std::vector<float> img;
    int r = 0, c = 0;
    const bool ok = colour ? gipuma_host::read_image_colour(path, img, r, c) : gipuma_host::read_image_gray(path, img, r, c);
    if (!ok) return -1;
    *rows = r;
    *cols = c;
    if (out) memcpy(out, img.data(), img.size() * sizeof(float));
    return 0;
}

int gipuma_host_main(int argc, char **argv)
{  // main(), main.cpp:1201-1230
    gipuma_host::InputFiles in;
    gipuma_host::OutputFiles out;
    gipuma_host::AlgorithmParameters ap;
    gipuma_host::GTcheckParameters gt;
    if (gipuma_host::parse_command_line(argc, argv, in, out, ap, &gt) < 0) return 1;
    std::string folder;
    const int rc = gipuma_host::run_gipuma(in, out, ap, &folder, &gt);
    if (!rc) std::cout << "Results written to " << folder << std::endl;
    return rc ? 1 : 0;
}
}
