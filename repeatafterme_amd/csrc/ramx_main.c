/* RAMExtend -- the CLI executable; all logic lives in libramx.so (ramx_cli.c). */
#include "ramx.h"
int main(int argc, char **argv) { return ramx_cli_main(argc, argv); }
