/* jni.h -- NOT a JNI implementation and not a substitute for the JDK's header: the handful of JNI 1.6 type and
 * function-table declarations that integration/jni/rm_jni.c uses, with the signatures the Java Native Interface
 * specification publishes, so that the glue can at least go through a C compiler's type checker in an image without a
 * JDK (tests/test_abi.py::test_jni_glue_type_checks).  Nothing links against this; a real build uses $JAVA_HOME/include. */
#ifndef RM_JNI_TYPECHECK_H
#define RM_JNI_TYPECHECK_H
#include <stdint.h>

typedef int32_t jint;
typedef int64_t jlong;
typedef int8_t jbyte;
typedef uint8_t jboolean;
typedef double jdouble;
typedef jint jsize;
struct _jobject;
typedef struct _jobject *jobject;
typedef jobject jclass;
typedef jobject jstring;
typedef jobject jarray;
typedef jarray jobjectArray;
typedef jarray jbyteArray;
typedef jarray jintArray;
typedef jarray jlongArray;
typedef jarray jdoubleArray;

#define JNIEXPORT __attribute__((visibility("default")))
#define JNICALL
#define JNI_ABORT 2

struct JNINativeInterface_;
typedef const struct JNINativeInterface_ *JNIEnv;
struct JNINativeInterface_ {
    jstring (*NewStringUTF)(JNIEnv *env, const char *utf);
    jsize (*GetArrayLength)(JNIEnv *env, jarray array);
    void (*SetObjectArrayElement)(JNIEnv *env, jobjectArray array, jsize index, jobject val);
    jbyte *(*GetByteArrayElements)(JNIEnv *env, jbyteArray array, jboolean *isCopy);
    jint *(*GetIntArrayElements)(JNIEnv *env, jintArray array, jboolean *isCopy);
    jdouble *(*GetDoubleArrayElements)(JNIEnv *env, jdoubleArray array, jboolean *isCopy);
    void (*ReleaseByteArrayElements)(JNIEnv *env, jbyteArray array, jbyte *elems, jint mode);
    void (*ReleaseIntArrayElements)(JNIEnv *env, jintArray array, jint *elems, jint mode);
    void (*ReleaseDoubleArrayElements)(JNIEnv *env, jdoubleArray array, jdouble *elems, jint mode);
    void (*SetIntArrayRegion)(JNIEnv *env, jintArray array, jsize start, jsize len, const jint *buf);
    void (*SetLongArrayRegion)(JNIEnv *env, jlongArray array, jsize start, jsize len, const jlong *buf);
    jobject (*NewDirectByteBuffer)(JNIEnv *env, void *address, jlong capacity);
};
#endif
