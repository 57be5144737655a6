/* san_driver.c -- TEST INFRASTRUCTURE: runs the CPU restatement (engine_ref.c) on op-stream files, Regular and Lockable
 * datablocks, as a plain executable so that it can be built with -fsanitize=address,undefined
 * (tests/test_oracle_sanitizers.py).  usage: san_driver <stream.ops>... */
#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>
#include "ocr_opstream.h"
int ocrref_run2(const ocrops_hdr_t *h, const ocrops_edt_t *edt, const ocrops_evt_t *evt, const uint64_t *db_init,
                const ocrops_op_t *main_ops, const ocrops_op_t *script,
                int32_t *level, uint64_t *val, uint64_t *evt_state, uint64_t *db_val, uint64_t *stats, uint32_t flags);
int main(int argc, char **argv) {
    for (int a = 1; a < argc; ++a) for (uint32_t fl = 0; fl < 2; ++fl) {
        FILE *f = fopen(argv[a], "rb"); ocrops_hdr_t h; if (!f || fread(&h, sizeof h, 1, f) != 1) return 2;
        ocrops_edt_t *edt = malloc(sizeof *edt * (h.n_edt + 1)); ocrops_evt_t *evt = malloc(sizeof *evt * (h.n_evt + 1));
        uint64_t *dbi = malloc(8 * (h.n_db + 1)); ocrops_op_t *m = malloc(sizeof *m * (h.n_main + 1)), *sc = malloc(sizeof *sc * (h.n_script + 1));
        if (fread(edt, sizeof *edt, h.n_edt, f) != h.n_edt || fread(evt, sizeof *evt, h.n_evt, f) != h.n_evt || fread(dbi, 8, h.n_db, f) != h.n_db ||
            fread(m, sizeof *m, h.n_main, f) != h.n_main || fread(sc, sizeof *sc, h.n_script, f) != h.n_script) return 3;
        fclose(f);
        int32_t *lv = malloc(4 * (h.n_edt + 1)); uint64_t *val = malloc(8 * (h.n_edt + 1)), *es = malloc(8 * (h.n_evt + h.n_edt + 1)), *dv = malloc(8 * (h.n_db + h.n_edt + 1)), st[16];
        ocrref_run2(&h, edt, evt, dbi, m, sc, lv, val, es, dv, st, fl);
        printf("%s flags %u levels %llu fired %llu\n", argv[a], fl, (unsigned long long)st[0], (unsigned long long)st[1]);
        free(edt); free(evt); free(dbi); free(m); free(sc); free(lv); free(val); free(es); free(dv);
    }
    return 0;
}
