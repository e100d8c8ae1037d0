/*
 * rm_jni.c -- thin JNI glue between se.sics.emul8.radiomedium.GpuRadioMedium
 * (integration/java/...) and the C ABI of libradiomedium_hip.so (include/radiomedium_hip.h).
 * Pure marshalling: every native method is one ABI call.
 *
 * Compiled only where a JDK exists (none in the build image):
 *   gcc -shared -fPIC -I$JAVA_HOME/include -I$JAVA_HOME/include/linux -I../../include \
 *       rm_jni.c -L../../radio-sim_amd/csrc -lradiomedium_hip -o libradiomedium_jni.so
 */
#if defined(__has_include)
#if __has_include(<jni.h>)
#define RM_HAVE_JNI 1
#endif
#endif

#ifdef RM_HAVE_JNI
#include <jni.h>
#include <stdint.h>

#include "radiomedium_hip.h"

#define JFN(name) JNICALL Java_se_sics_emul8_radiomedium_GpuRadioMedium_##name

JNIEXPORT jlong JFN(nCreate)(JNIEnv *env, jclass cls, jint device)
{
    rm_context *ctx = NULL;
    (void)env; (void)cls;
    return rm_create(device, &ctx) == RM_OK ? (jlong)(intptr_t)ctx : 0;
}

JNIEXPORT void JFN(nDestroy)(JNIEnv *env, jclass cls, jlong ctx)
{
    (void)env; (void)cls;
    rm_destroy((rm_context *)(intptr_t)ctx);
}

JNIEXPORT jstring JFN(nLastError)(JNIEnv *env, jclass cls)
{
    (void)cls;
    return (*env)->NewStringUTF(env, rm_last_error());
}

JNIEXPORT jstring JFN(nGetName)(JNIEnv *env, jclass cls, jlong ctx)
{
    (void)cls;
    return (*env)->NewStringUTF(env, rm_get_name((rm_context *)(intptr_t)ctx));
}

JNIEXPORT jint JFN(nSetModel)(JNIEnv *env, jclass cls, jlong ctx, jint kind, jint flags, jdoubleArray params)
{
    rm_model_params p;
    (void)cls;
    rm_model_defaults(&p, kind);
    p.flags = flags;
    if (params != NULL) { /* order: the double fields of rm_model_params from udgm_success_ratio_tx on */
        jsize n = (*env)->GetArrayLength(env, params);
        jdouble *v = (*env)->GetDoubleArrayElements(env, params, NULL);
        double *f[] = {&p.udgm_success_ratio_tx, &p.udgm_success_ratio_rx, &p.udgm_transmission_range,
                       &p.udgm_interference_range, &p.const_range, &p.ld_pl0_db, &p.ld_exponent, &p.ld_d0,
                       &p.ld_sigma_db, &p.ld_clip, &p.ld_sensitivity_dbm, &p.ld_noise_dbm, &p.ld_capture_db,
                       &p.ld_ifloor_dbm};
        for (jsize i = 0; i < n && i < (jsize)(sizeof(f) / sizeof(f[0])); i++) *f[i] = v[i];
        (*env)->ReleaseDoubleArrayElements(env, params, v, JNI_ABORT);
    }
    return rm_set_model((rm_context *)(intptr_t)ctx, &p);
}

JNIEXPORT jint JFN(nSetN2NMatrix)(JNIEnv *env, jclass cls, jlong ctx, jint m, jdoubleArray rows)
{
    (void)cls;
    jdouble *v = (*env)->GetDoubleArrayElements(env, rows, NULL);
    int rc = rm_set_n2n_matrix((rm_context *)(intptr_t)ctx, m, v);
    (*env)->ReleaseDoubleArrayElements(env, rows, v, JNI_ABORT);
    return rc;
}

JNIEXPORT jint JFN(nSeed)(JNIEnv *env, jclass cls, jlong ctx, jlong seed)
{
    (void)env; (void)cls;
    return rm_seed((rm_context *)(intptr_t)ctx, seed);
}

JNIEXPORT jint JFN(nSetTime)(JNIEnv *env, jclass cls, jlong ctx, jlong t)
{
    (void)env; (void)cls;
    return rm_set_time((rm_context *)(intptr_t)ctx, t);
}

JNIEXPORT jint JFN(nNodesUpload)(JNIEnv *env, jclass cls, jlong ctx, jint n, jdoubleArray x, jdoubleArray y,
                                 jdoubleArray z, jdoubleArray txpower, jintArray channel, jbyteArray enabled,
                                 jdoubleArray rxprob, jdoubleArray txprob, jintArray intId)
{
    (void)cls;
    jdouble *px = (*env)->GetDoubleArrayElements(env, x, NULL), *py = (*env)->GetDoubleArrayElements(env, y, NULL);
    jdouble *pz = (*env)->GetDoubleArrayElements(env, z, NULL), *pt = (*env)->GetDoubleArrayElements(env, txpower, NULL);
    jdouble *pr = (*env)->GetDoubleArrayElements(env, rxprob, NULL), *pp = (*env)->GetDoubleArrayElements(env, txprob, NULL);
    jint *pc = (*env)->GetIntArrayElements(env, channel, NULL), *pi = (*env)->GetIntArrayElements(env, intId, NULL);
    jbyte *pe = (*env)->GetByteArrayElements(env, enabled, NULL);
    int rc = rm_nodes_upload((rm_context *)(intptr_t)ctx, n, px, py, pz, pt, (const int32_t *)pc, (const uint8_t *)pe,
                             pr, pp, (const int32_t *)pi);
    (*env)->ReleaseDoubleArrayElements(env, x, px, JNI_ABORT);
    (*env)->ReleaseDoubleArrayElements(env, y, py, JNI_ABORT);
    (*env)->ReleaseDoubleArrayElements(env, z, pz, JNI_ABORT);
    (*env)->ReleaseDoubleArrayElements(env, txpower, pt, JNI_ABORT);
    (*env)->ReleaseDoubleArrayElements(env, rxprob, pr, JNI_ABORT);
    (*env)->ReleaseDoubleArrayElements(env, txprob, pp, JNI_ABORT);
    (*env)->ReleaseIntArrayElements(env, channel, pc, JNI_ABORT);
    (*env)->ReleaseIntArrayElements(env, intId, pi, JNI_ABORT);
    (*env)->ReleaseByteArrayElements(env, enabled, pe, JNI_ABORT);
    return rc;
}

JNIEXPORT jint JFN(nNodeUpdate)(JNIEnv *env, jclass cls, jlong ctx, jint node, jdouble x, jdouble y, jdouble z,
                                jdouble txpower, jint channel, jboolean enabled, jdouble rxprob, jdouble txprob)
{
    (void)env; (void)cls;
    return rm_node_update((rm_context *)(intptr_t)ctx, node, x, y, z, txpower, channel, enabled ? 1 : 0, rxprob, txprob);
}

JNIEXPORT jint JFN(nTransmit)(JNIEnv *env, jclass cls, jlong ctx, jint src, jlong startUs, jlong hexLength,
                              jboolean hasPower, jdouble txpower, jboolean hasChannel, jint channel, jintArray dst,
                              jbyteArray verdict, jdoubleArray rssi, jdoubleArray sinr, jbyteArray interference)
{
    (void)cls;
    jsize cap = (*env)->GetArrayLength(env, dst);
    jint *pd = (*env)->GetIntArrayElements(env, dst, NULL);
    jbyte *pv = (*env)->GetByteArrayElements(env, verdict, NULL);
    jdouble *pr = (*env)->GetDoubleArrayElements(env, rssi, NULL), *ps = (*env)->GetDoubleArrayElements(env, sinr, NULL);
    jbyte *pi = (*env)->GetByteArrayElements(env, interference, NULL);
    uint32_t count = 0;
    double tp = txpower;
    int32_t ch = channel;
    int rc = rm_transmit((rm_context *)(intptr_t)ctx, src, startUs, hexLength, hasPower ? &tp : NULL,
                         hasChannel ? &ch : NULL, (int32_t *)pd, (uint8_t *)pv, pr, ps, (uint32_t)cap, &count,
                         (uint8_t *)pi);
    (*env)->ReleaseIntArrayElements(env, dst, pd, 0);
    (*env)->ReleaseByteArrayElements(env, verdict, pv, 0);
    (*env)->ReleaseDoubleArrayElements(env, rssi, pr, 0);
    (*env)->ReleaseDoubleArrayElements(env, sinr, ps, 0);
    (*env)->ReleaseByteArrayElements(env, interference, pi, 0);
    return rc == RM_OK ? (jint)count : (jint)rc;
}

/* ---- tick mode ------------------------------------------------------------------------------------ */
JNIEXPORT jint JFN(nTickBegin)(JNIEnv *env, jclass cls, jlong ctx, jlong t0, jlong t1)
{
    (void)env; (void)cls;
    return rm_tick_begin((rm_context *)(intptr_t)ctx, t0, t1);
}

JNIEXPORT jint JFN(nEnqueue)(JNIEnv *env, jclass cls, jlong ctx, jint src, jlong startUs, jlong airUs, jdouble txpower, jint channel)
{
    (void)env; (void)cls;
    double tp = txpower;
    int32_t ch = channel;
    return rm_enqueue_tx((rm_context *)(intptr_t)ctx, src, startUs, airUs, &tp, &ch);
}

JNIEXPORT jint JFN(nTickRun)(JNIEnv *env, jclass cls, jlong ctx)
{
    (void)env; (void)cls;
    return rm_tick_run((rm_context *)(intptr_t)ctx);
}

/* the result is not copied: direct ByteBuffers over the context's pinned, host-mapped block */
JNIEXPORT jint JFN(nTickFlushView)(JNIEnv *env, jclass cls, jlong ctx, jobjectArray views, jintArray counts)
{
    (void)cls;
    rm_host_result r;
    int rc = rm_tick_flush_view((rm_context *)(intptr_t)ctx, &r);
    if (rc != RM_OK) return rc;
    jint c[2] = {(jint)r.count, (jint)r.n_packets};
    (*env)->SetIntArrayRegion(env, counts, 0, 2, c);
    /* (ABI version 5) views[4]: the links' rssi -- or, for the reference's media, NULL and views[5] the PACKETS' rssi */
    void *ptr[6] = {(void *)r.pkt_offset, (void *)r.pkt_interference, (void *)r.dst, (void *)r.verdict, (void *)r.rssi, (void *)r.pkt_rssi};
    jlong len[6] = {((jlong)r.n_packets + 1) * 4, (jlong)r.n_packets, (jlong)r.count * 4, (jlong)r.count, r.rssi ? (jlong)r.count * 8 : 0,
                    r.pkt_rssi ? (jlong)r.n_packets * 8 : 0};
    for (int i = 0; i < 6; i++)
        (*env)->SetObjectArrayElement(env, views, i, (*env)->NewDirectByteBuffer(env, ptr[i], len[i] > 0 ? len[i] : 0));
    return RM_OK;
}

JNIEXPORT jint JFN(nAbiVersion)(JNIEnv *env, jclass cls)
{
    (void)env; (void)cls;
    return rm_abi_version() == RM_ABI_VERSION ? RM_ABI_VERSION : -rm_abi_version(); /* header and library must agree too */
}

/* ---- reception stage on the device ------------------------------------------------------------------ */
JNIEXPORT jint JFN(nEventsEnable)(JNIEnv *env, jclass cls, jlong ctx, jint maxPackets, jint maxLinks)
{
    (void)env; (void)cls;
    if (maxPackets == 0 && maxLinks == 0) return rm_events_disable((rm_context *)(intptr_t)ctx);
    return rm_events_enable((rm_context *)(intptr_t)ctx, (uint32_t)maxPackets, (uint32_t)maxLinks);
}

JNIEXPORT jlong JFN(nEventsNextPacket)(JNIEnv *env, jclass cls, jlong ctx)
{
    (void)env; (void)cls;
    return rm_events_next_packet((rm_context *)(intptr_t)ctx);
}

JNIEXPORT jint JFN(nEventsProcess)(JNIEnv *env, jclass cls, jlong ctx, jlong timeUs, jobjectArray views, jlongArray counts)
{
    (void)cls;
    rm_delivery_view v;
    int rc = rm_events_process((rm_context *)(intptr_t)ctx, timeUs, &v);
    if (rc != RM_OK) return rc;
    jlong c[4] = {(jlong)v.count, (jlong)v.pending_packets, (jlong)v.oldest_packet, (jlong)v.n_runs};
    (*env)->SetLongArrayRegion(env, counts, 0, 4, c);
    /* the packet numbers once per run of deliveries (a packet's deliveries are adjacent), then the deliveries */
    void *ptr[5] = {(void *)v.run_packet, (void *)v.run_first, (void *)v.run_count, (void *)v.dst, (void *)v.rssi};
    jlong len[5] = {(jlong)v.n_runs * 8, (jlong)v.n_runs * 4, (jlong)v.n_runs * 4, (jlong)v.count * 4, (jlong)v.count * 8};
    for (int i = 0; i < 5; i++) (*env)->SetObjectArrayElement(env, views, i, (*env)->NewDirectByteBuffer(env, ptr[i], len[i]));
    return RM_OK;
}

JNIEXPORT jint JFN(nNodeInfo)(JNIEnv *env, jclass cls, jlong ctx, jintArray nodes, jdoubleArray rssi, jintArray receiving,
                              jintArray channel)
{
    (void)cls;
    jsize n = (*env)->GetArrayLength(env, nodes);
    jint *pn = (*env)->GetIntArrayElements(env, nodes, NULL);
    jdouble *pr = (*env)->GetDoubleArrayElements(env, rssi, NULL);
    jint *px = (*env)->GetIntArrayElements(env, receiving, NULL), *pc = (*env)->GetIntArrayElements(env, channel, NULL);
    int rc = rm_node_info((rm_context *)(intptr_t)ctx, (const int32_t *)pn, n, pr, (int32_t *)px, (int32_t *)pc);
    (*env)->ReleaseIntArrayElements(env, nodes, pn, JNI_ABORT);
    (*env)->ReleaseDoubleArrayElements(env, rssi, pr, 0);
    (*env)->ReleaseIntArrayElements(env, receiving, px, 0);
    (*env)->ReleaseIntArrayElements(env, channel, pc, 0);
    return rc;
}

JNIEXPORT jint JFN(nNodeInfoChanged)(JNIEnv *env, jclass cls, jlong ctx, jintArray nodes, jdoubleArray rssi, jintArray receiving,
                                     jintArray channel)
{
    (void)cls;
    jsize cap = (*env)->GetArrayLength(env, nodes);
    jint *pn = (*env)->GetIntArrayElements(env, nodes, NULL);
    jdouble *pr = (*env)->GetDoubleArrayElements(env, rssi, NULL);
    jint *px = (*env)->GetIntArrayElements(env, receiving, NULL), *pc = (*env)->GetIntArrayElements(env, channel, NULL);
    int32_t count = 0;
    int rc = rm_node_info_changed((rm_context *)(intptr_t)ctx, (int32_t *)pn, pr, (int32_t *)px, (int32_t *)pc, cap, &count);
    (*env)->ReleaseIntArrayElements(env, nodes, pn, 0);
    (*env)->ReleaseDoubleArrayElements(env, rssi, pr, 0);
    (*env)->ReleaseIntArrayElements(env, receiving, px, 0);
    (*env)->ReleaseIntArrayElements(env, channel, pc, 0);
    return rc == RM_OK ? count : -1;
}
#else
/* no JDK on this machine: nothing to build (the C ABI is exercised through ctypes and C++ instead) */
typedef int rm_jni_unavailable;
#endif
