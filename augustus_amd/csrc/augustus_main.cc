// augustus_main.cc -- the `augustus` executable of the MI355X path: same command line as the reference binary
// (reference src/augustus.cc:94-248), implemented by augx_main() in libaugx.
#include "../../include/augx.h"
int main(int argc, char **argv) { return augx_main(argc, (const char *const *)argv); }
