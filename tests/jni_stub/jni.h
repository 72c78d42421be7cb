/* Minimal stand-in for <jni.h> (TEST INFRASTRUCTURE): just enough of the JNI surface for bindings/jni/cookmatch_jni.c to
 * be type-checked against include/cookmatch.h in an image without a JDK.  Never shipped. */
#ifndef JNI_STUB_H
#define JNI_STUB_H
#include <stdint.h>
typedef int32_t jint;
typedef int64_t jlong;
typedef int32_t jsize;
typedef uint8_t jboolean;
typedef void* jobject;
typedef jobject jclass;
typedef jobject jstring;
typedef jobject jobjectArray;
#define JNIEXPORT __attribute__((visibility("default")))
#define JNICALL
struct JNINativeInterface_;
typedef const struct JNINativeInterface_* JNIEnv;
struct JNINativeInterface_ {
  void* (*GetDirectBufferAddress)(JNIEnv*, jobject);
  jsize (*GetArrayLength)(JNIEnv*, jobjectArray);
  jobject (*GetObjectArrayElement)(JNIEnv*, jobjectArray, jsize);
  jstring (*NewStringUTF)(JNIEnv*, const char*);
  jlong (*GetDirectBufferCapacity)(JNIEnv*, jobject);
  void (*DeleteLocalRef)(JNIEnv*, jobject);
  jobject (*NewDirectByteBuffer)(JNIEnv*, void*, jlong);
};
#endif
