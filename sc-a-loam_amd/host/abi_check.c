/* Plain C99 caller of libscaloam_hip.so: include/scaloam_hip.h must compile with `gcc -std=c99 -pedantic-errors` and every stage's
 * create entry point must link and fail cleanly (SCAL_E_NO_DEVICE) on a host without a gfx950 device, or succeed and be destroyed on one.
 * Prints "abi ok <version> devices=<n>"; exit code 0. */
#include <stdio.h>
#include <string.h>
#include "scaloam_hip.h"

int main(void) {
    scal_features_config fc;
    scal_features_t* f = NULL;
    scal_pipeline_config pc;
    scal_pipeline_t* p = NULL;
    int rc, n = scal_device_count();
    memset(&fc, 0, sizeof fc);
    fc.lidar_type = SCAL_HDL64, fc.n_scans = 64, fc.minimum_range = 5.0, fc.max_points = 1000, fc.check_finite = 1;
    rc = scal_features_create(&fc, &f);
    if (n < 1 && rc != SCAL_E_NO_DEVICE) {
        printf("expected SCAL_E_NO_DEVICE without a GPU, got %d (%s)\n", rc, scal_last_error());
        return 1;
    }
    if (rc == SCAL_OK) scal_features_destroy(f);
    memset(&pc, 0, sizeof pc);
    pc.lidar_type = SCAL_HDL64, pc.n_scans = 64, pc.minimum_range = 5.0, pc.max_points = 1000, pc.line_res = 0.4f, pc.plane_res = 0.8f;
    pc.max_map_points = 100000, pc.ring = 99;
    if (scal_pipeline_create(&pc, &p) != SCAL_E_ARG) { /* argument validation comes before any device work */
        printf("scal_pipeline_create accepted ring = 99\n");
        return 1;
    }
    printf("abi ok %s devices=%d\n", scal_version(), n);
    return 0;
}
