// dbtk_ktools.cpp — `ktools serialize PREF` over the C-ABI (include/dbtk.h: dbtk_rpgg_serialize), the one ktools
// subcommand the align path depends on (src/kmertools.cpp:221-345).  Same usage text and exit behaviour for that
// subcommand; the others are not part of this repository.
#include <stdio.h>
#include <string.h>

#include "../../include/dbtk.h"

int main(int argc, char* argv[]) {
    if (argc < 2 || strcmp(argv[1], "serialize") != 0) {
        fprintf(stderr, "Usage: ktools serialize <pref>\n\n  (only `serialize` is provided here)\n");
        return argc < 2 ? 0 : 1;
    }
    if (argc == 2) {
        fprintf(stderr, "Usage: ktools serialize <pref>\n\n  PREF     prefix of *.(graph|fl|tr).kmers\n");
        return 0;
    }
    if (dbtk_rpgg_serialize(argv[2]) != DBTK_OK) {
        fprintf(stderr, "ktools: %s\n", dbtk_last_error());
        return 134;  // the reference asserts on unusable files
    }
    fprintf(stderr, "done\n");
    return 0;
}
