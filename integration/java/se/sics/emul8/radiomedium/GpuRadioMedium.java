/*
 * GpuRadioMedium -- the reference-side binding a radio-sim maintainer adds to attach the MI355X
 * engine (libradiomedium_hip.so) behind the existing RadioMedium plug-in contract.
 *
 * Build-owned file (not a copy of any reference source).  It lives in the reference's package so
 * that it can extend AbstractRadioMedium (AbstractRadioMedium.java:35-53) and use the
 * package-visible Simulator / Node / Transciever accessors.  NOT compiled in this repository:
 * there is no JDK in the build image (SURVEY.md section 0.4).
 *
 * Semantics mirrored (paths relative to radio-medium/java/se/sics/emul8/radiomedium/):
 *   transmit(RadioPacket)   UDGMRadioMedium.java:83-117 and siblings: one native call, then the
 *                           heard links come back in node order and are turned into exactly the
 *                           calls the reference makes (generateTransmissionEvents once,
 *                           generateReceptionEvents per heard link, or deliverRadioPacket for the
 *                           constant-loss medium, UDGMConstantLossRadioMedium.java:31).
 *   node state              the reference has no change notification (SURVEY.md section 3.3):
 *                           the node table is re-uploaded when its length or a version stamp
 *                           changed; positions / radio fields are read through the public getters.
 *   threading               transmit() is entered from per-socket reader threads
 *                           (net/JSONClientConnection.java:118-131): one lock per context.
 *   errors                  transmit() returns void and must not throw: failures are logged and
 *                           the packet reaches no receiver.
 */
package se.sics.emul8.radiomedium;

import org.slf4j.Logger;
import org.slf4j.LoggerFactory;

public class GpuRadioMedium extends AbstractRadioMedium {

    private static final Logger log = LoggerFactory.getLogger(GpuRadioMedium.class);

    public static final int MODEL_NULL = 0, MODEL_UDGM = 1, MODEL_UDGM_CONST = 2, MODEL_N2N = 3, MODEL_LOGDIST = 4;
    private static final byte INTERFERED = 1, DELIVERED = 2;

    static {
        System.loadLibrary("radiomedium_jni"); // integration/jni/rm_jni.c, links libradiomedium_hip.so
    }

    /* ---- native side: one-to-one with include/radiomedium_hip.h ---- */
    private static native long nCreate(int device);
    private static native void nDestroy(long ctx);
    private static native String nLastError();
    private static native String nGetName(long ctx);
    private static native int nSetModel(long ctx, int kind, int flags, double[] params);
    private static native int nSetN2NMatrix(long ctx, int m, double[] rowMajor);
    private static native int nSeed(long ctx, long seed);
    private static native int nNodesUpload(long ctx, int n, double[] x, double[] y, double[] z, double[] txpower,
            int[] channel, byte[] enabled, double[] rxprob, double[] txprob, int[] intId);
    private static native int nNodeUpdate(long ctx, int node, double x, double y, double z, double txpower, int channel,
            boolean enabled, double rxprob, double txprob);
    private static native int nSetTime(long ctx, long currentTime);
    /** returns the number of heard links (negative = rm error); dst/verdict/rssi are filled in node order */
    private static native int nTransmit(long ctx, int src, long startUs, long hexLength, boolean hasPower, double txpower,
            boolean hasChannel, int channel, int[] dst, byte[] verdict, double[] rssi, double[] sinr, byte[] interference);

    private final Object lock = new Object();
    private final int kind;
    private long ctx;
    private Node[] uploaded;          // the Simulator.getNodes() snapshot the device currently mirrors
    private java.util.IdentityHashMap<Node, Integer> index = new java.util.IdentityHashMap<Node, Integer>();
    private final java.util.ArrayList<Node> changed = new java.util.ArrayList<Node>(); // the dirty list
    private int[] dst = new int[0];
    private byte[] verdict = new byte[0];
    private double[] rssi = new double[0], sinr = new double[0];

    public GpuRadioMedium(int kind, long randomSeed) {
        this.kind = kind;
        this.ctx = nCreate(0);
        if (this.ctx == 0) {
            throw new IllegalStateException("no MI355X radio medium: " + nLastError());
        }
        nSetModel(ctx, kind, 0, null);
        nSeed(ctx, randomSeed); // a seeded Simulator(Random) (Simulator.java:87-89) must use the same seed
    }

    @Override
    public String getName() {
        return nGetName(ctx) + " [MI355X]";
    }

    private void syncNodes(Node[] nodes) {
        if (nodes == uploaded) { // copy-on-write array (Simulator.java:274): same array == same node set
            for (Node nd : changed) { // the dirty list: rm_node_update writes these nodes in place on the device
                Integer i = index.get(nd);
                Transciever r = nd.getRadio();
                if (i != null && nNodeUpdate(ctx, i, nd.getPosition().x, nd.getPosition().y, nd.getPosition().z,
                        r.getTransmitPower(), r.getWirelessChannel(), r.isEnabled(), r.getRxProbability(),
                        r.getTxProbability()) != 0) {
                    log.error("node update failed: {}", nLastError());
                }
            }
            changed.clear();
            return;
        }
        changed.clear(); // covered by the snapshot below
        int n = nodes.length;
        double[] x = new double[n], y = new double[n], z = new double[n], tp = new double[n], rp = new double[n], xp = new double[n];
        int[] ch = new int[n], id = new int[n];
        byte[] en = new byte[n];
        index.clear();
        for (int i = 0; i < n; i++) {
            Node nd = nodes[i];
            Transciever r = nd.getRadio();
            x[i] = nd.getPosition().x; y[i] = nd.getPosition().y; z[i] = nd.getPosition().z;
            tp[i] = r.getTransmitPower(); ch[i] = r.getWirelessChannel(); en[i] = (byte) (r.isEnabled() ? 1 : 0);
            rp[i] = r.getRxProbability(); xp[i] = r.getTxProbability(); id[i] = nd.getIdAsInteger();
            index.put(nd, i);
        }
        if (nNodesUpload(ctx, n, x, y, z, tp, ch, en, rp, xp, id) != 0) {
            log.error("node upload failed: {}", nLastError());
        }
        uploaded = nodes;
        if (dst.length < n) {
            dst = new int[n]; verdict = new byte[n]; rssi = new double[n]; sinr = new double[n];
        }
    }

    /** call after node-config-set changed fields of an existing node (no hook exists in the reference,
     *  SimulatorJSONHandler.java:105-143): only this node is written to the device before the next packet */
    public void nodeChanged(Node node) {
        synchronized (lock) {
            changed.add(node);
        }
    }

    /** anything may have changed: the next packet uploads a fresh snapshot */
    public void invalidateNodes() {
        synchronized (lock) {
            uploaded = null;
        }
    }

    @Override
    public void transmit(RadioPacket packet) {
        Simulator sim = this.simulator;
        if (sim == null) {
            log.error("No simulator"); // NullRadioMedium.java:49-53
            return;
        }
        Node[] nodes = sim.getNodes();
        if (nodes == null) {
            return;
        }
        synchronized (lock) {
            syncNodes(nodes);
            Integer src = index.get(packet.getSource());
            if (src == null) {
                log.error("source node not registered");
                return;
            }
            nSetTime(ctx, sim.getTime());
            byte[] interference = new byte[1];
            int heard = nTransmit(ctx, src, packet.getStartTime(), packet.getPacketDataAsHex() == null ? 0
                    : packet.getPacketDataAsHex().length(), true, packet.getTransmitPower(), true,
                    packet.getWirelessChannel(), dst, verdict, rssi, sinr, interference);
            if (heard < 0) {
                log.error("radio medium: {}", nLastError());
                return;
            }
            if (kind != MODEL_UDGM_CONST) {
                sim.generateTransmissionEvents(packet); // UDGMRadioMedium.java:97
            }
            for (int i = 0; i < heard; i++) { // node order, as the reference's loop (:99)
                Node node = nodes[dst[i]];
                if (kind == MODEL_UDGM_CONST) {
                    sim.deliverRadioPacket(packet, node, rssi[i]); // immediate delivery, no events
                } else {
                    sim.generateReceptionEvents(packet, node, rssi[i], verdict[i] == DELIVERED);
                }
            }
        }
    }

    public void close() {
        synchronized (lock) {
            if (ctx != 0) {
                nDestroy(ctx);
                ctx = 0;
            }
        }
    }
}
