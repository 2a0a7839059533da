/*
 * hostref_harness.cpp -- runs the reference's OWN camera front-end on the CPU and prints what it produces.
 * TEST INFRASTRUCTURE.  build_hostref.sh pipes, untouched and never written to disk:
 *     /root/reference/main.h, algorithmparameters.h, cameraparameters.h (included where they lie),
 *     fileIoUtils.h lines 1..112 (the .P / KITTI / bounding-volume readers),
 *     cameraGeometryUtils.h (whole: getCameraParameters :174-353 and its helpers),
 *     main.cpp lines 430..499 (selectViews)
 * between the CUDA-on-CPU shim (managed memory = calloc), the functional mini OpenCV (opencv2/cv_mini.hpp) and
 * this file.  Everything the reference computes between reading the .P files and handing Camera_cu /
 * viewSelectionSubset / the depth range to the path (main.cpp:686-912) is therefore the reference's code.
 *
 *   hostref <p_folder/> <cam_scale> <cols> <rows> <min_angle> <max_angle> <max_views> <depth_min> <depth_max> img0 img1 ...
 * reads <p_folder><imgK>.P for every image name (cameraGeometryUtils.h:219-224), image 0 = the reference view,
 * and prints one JSON object.
 */
int main(int argc, char **argv)
{
    if (argc < 11) {
        fprintf(stderr, "usage: hostref p_folder cam_scale cols rows min_angle max_angle max_views depth_min depth_max img...\n");
        return 2;
    }
    InputFiles inputFiles;
    inputFiles.p_folder = argv[1];
    const float cam_scale = (float)atof(argv[2]);
    const int cols = atoi(argv[3]), rows = atoi(argv[4]);
    AlgorithmParameters *algParams = new AlgorithmParameters;
    algParams->min_angle = (float)atof(argv[5]);
    algParams->max_angle = (float)atof(argv[6]);
    algParams->max_views = (unsigned)atoi(argv[7]);
    algParams->depthMin = (float)atof(argv[8]);
    algParams->depthMax = (float)atof(argv[9]);
    for (int i = 10; i < argc; i++) inputFiles.img_filenames.push_back(argv[i]);
    const size_t n = inputFiles.img_filenames.size();
    if (n > MAX_IMAGES) return 3;
    CameraParameters_cu *cpc = new CameraParameters_cu;
    /* main.cpp:795 */
    CameraParameters cameraParams = getCameraParameters(*cpc, inputFiles, cam_scale);
    /* main.cpp:870 */
    selectViews(cameraParams, cols, rows, *algParams);
    /* main.cpp:905-906 */
    const float min_disp = disparityDepthConversion(cameraParams.f, cameraParams.cameras[0].baseline, algParams->depthMax);
    const float max_disp = disparityDepthConversion(cameraParams.f, cameraParams.cameras[0].baseline, algParams->depthMin);

    auto arr = [](const char *name, const float *v, int k, bool comma = true) {
        printf("\"%s\": [", name);
        for (int i = 0; i < k; i++) printf("%s%.9g", i ? ", " : "", (double)v[i]);
        printf("]%s", comma ? ", " : "");
    };
    printf("{\"cameras\": [");
    for (size_t i = 0; i < n; i++) {
        const Camera_cu &c = cpc->cameras[i];
        printf("%s{", i ? ", " : "");
        arr("K", c.K, 9); arr("K_inv", c.K_inv, 9); arr("R", c.R, 9); arr("M_inv", c.M_inv, 9);
        arr("R_orig_inv", c.R_orig_inv, 9); arr("P", c.P, 12);
        const float t[3] = {c.t4.x, c.t4.y, c.t4.z}, C[3] = {c.C4.x, c.C4.y, c.C4.z};
        const float p34[3] = {c.P_col34.x, c.P_col34.y, c.P_col34.z};
        arr("t", t, 3); arr("C", C, 3); arr("P_col34", p34, 3);
        printf("\"fx\": %.9g, \"fy\": %.9g, \"f\": %.9g, \"alpha\": %.9g, \"baseline\": %.9g}", (double)c.fx, (double)c.fy,
               (double)c.f, (double)c.alpha, (double)c.baseline);
    }
    printf("], \"f\": %.9g, \"subset\": [", (double)cpc->f);
    for (size_t i = 0; i < cameraParams.viewSelectionSubset.size(); i++)
        printf("%s%d", i ? ", " : "", cameraParams.viewSelectionSubset[i]);
    printf("], \"depth_min\": %.9g, \"depth_max\": %.9g, \"min_disparity\": %.9g, \"max_disparity\": %.9g}\n",
           (double)algParams->depthMin, (double)algParams->depthMax, (double)min_disp, (double)max_disp);
    return 0;
}
