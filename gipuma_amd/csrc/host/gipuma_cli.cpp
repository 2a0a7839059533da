// gipuma_cli.cpp -- `gipuma_hip <images...> -images_folder ... -p_folder ... --blocksize=...`:
// the reference's command-line surface (main.cpp:164-428) on the MI355X path.
#include "gipuma_host.h"

int main(int argc, char **argv) { return gipuma_host_main(argc, argv); }
