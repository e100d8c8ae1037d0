/* udgm_transmit.c -- the C ABI from plain C: the reference's UDGM medium, four nodes, one packet.
 *
 *   gcc -std=c99 -Iinclude examples/udgm_transmit.c -Lradio-sim_amd/csrc -lradiomedium_hip \
 *       -Wl,-rpath,$PWD/radio-sim_amd/csrc -o udgm_transmit && ./udgm_transmit
 *
 * Node 0 transmits; node 1 sits exactly on the transmission range (50 m: heard, d == range is in
 * range, UDGMRadioMedium.java:74-77), node 2 one millimetre beyond (unheard), node 3 is on
 * another channel (unheard).  Prints the heard receivers in node order, as the reference's loop
 * would call generateReceptionEvents for them (UDGMRadioMedium.java:99-115).
 */
#include <stdio.h>

#include "radiomedium_hip.h"

int main(void)
{
    rm_context *ctx = NULL;
    if (rm_create(0, &ctx) != RM_OK) {
        fprintf(stderr, "rm_create: %s\n", rm_last_error());
        return 2; /* no gfx950 device: there is no CPU fallback */
    }
    const double x[4] = {0.0, 30.0, 30.0, 10.0}, y[4] = {0.0, 40.0, 40.0, 0.0}, z[4] = {0.0, 0.0, 0.001, 0.0};
    const int32_t channel[4] = {26, 26, 26, 25};
    rm_model_params p;
    rm_model_defaults(&p, RM_MODEL_UDGM);
    int rc = rm_nodes_upload(ctx, 4, x, y, z, NULL, channel, NULL, NULL, NULL, NULL);
    if (rc == RM_OK) rc = rm_set_model(ctx, &p);
    if (rc == RM_OK) rc = rm_seed(ctx, 42);
    int32_t dst[4];
    uint8_t verdict[4], interference = 0;
    double rssi[4], sinr[4];
    uint32_t heard = 0;
    /* the packet "0102030405" (10 hex characters = 320 us on the air) from node 0 at t = 1000 us */
    if (rc == RM_OK) rc = rm_transmit(ctx, 0, 1000, 10, NULL, NULL, dst, verdict, rssi, sinr, 4, &heard, &interference);
    if (rc != RM_OK) {
        fprintf(stderr, "error %d: %s\n", rc, rm_last_error());
        rm_destroy(ctx);
        return 1;
    }
    printf("%s: %u heard, Tx %s\n", rm_get_name(ctx), heard, interference ? "failed" : "ok");
    for (uint32_t i = 0; i < heard; ++i)
        printf("  node %d: %s, rssi %.1f\n", dst[i], verdict[i] == RM_DELIVERED ? "delivered" : "interfered", rssi[i]);
    rm_destroy(ctx);
    return (heard == 1 && dst[0] == 1 && verdict[0] == RM_DELIVERED) ? 0 : 1;
}
