#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f2 __attribute__((ext_vector_type(2)));
template <class T> __device__ __forceinline__ void cg_431(const T* xw, const float* y, T* acc) {
    { const T p = xw[0] * y[0];
      acc[2] += 4.082482905e-01f * p;
    }
    { const T p = xw[0] * y[6];
      acc[0] += 4.082482905e-01f * p;
    }
    { const T p = xw[1] * y[0];
      acc[1] += 2.886751346e-01f * p;
    }
    { const T p = xw[1] * y[1];
      acc[2] += 3.535533906e-01f * p;
    }
    { const T p = xw[1] * y[5];
      acc[0] += 3.535533906e-01f * p;
    }
    { const T p = xw[2] * y[0];
      acc[2] += -7.715167498e-02f * p;
    }
    { const T p = xw[2] * y[1];
      acc[1] += 3.779644730e-01f * p;
    }
    { const T p = xw[2] * y[2];
      acc[2] += 2.988071523e-01f * p;
    }
    { const T p = xw[2] * y[4];
      acc[0] += 2.988071523e-01f * p;
    }
    { const T p = xw[2] * y[6];
      acc[0] += 7.715167498e-02f * p;
    }
    { const T p = xw[3] * y[1];
      acc[2] += -1.336306210e-01f * p;
    }
    { const T p = xw[3] * y[2];
      acc[1] += 4.225771274e-01f * p;
    }
    { const T p = xw[3] * y[3];
      acc[0] += 3.450327797e-01f * p;
    }
    { const T p = xw[3] * y[5];
      acc[0] += 1.336306210e-01f * p;
    }
    { const T p = xw[4] * y[2];
      acc[0] += -2.672612419e-01f * p;
    }
    { const T p = xw[4] * y[3];
      acc[1] += 4.364357805e-01f * p;
    }
    { const T p = xw[4] * y[4];
      acc[2] += -2.672612419e-01f * p;
    }
    { const T p = xw[5] * y[1];
      acc[0] += -1.336306210e-01f * p;
    }
    { const T p = xw[5] * y[3];
      acc[2] += 3.450327797e-01f * p;
    }
    { const T p = xw[5] * y[4];
      acc[1] += 4.225771274e-01f * p;
    }
    { const T p = xw[5] * y[5];
      acc[2] += -1.336306210e-01f * p;
    }
    { const T p = xw[6] * y[0];
      acc[0] += -7.715167498e-02f * p;
    }
    { const T p = xw[6] * y[2];
      acc[0] += -2.988071523e-01f * p;
    }
    { const T p = xw[6] * y[4];
      acc[2] += 2.988071523e-01f * p;
    }
    { const T p = xw[6] * y[5];
      acc[1] += 3.779644730e-01f * p;
    }
    { const T p = xw[6] * y[6];
      acc[2] += -7.715167498e-02f * p;
    }
    { const T p = xw[7] * y[1];
      acc[0] += -3.535533906e-01f * p;
    }
    { const T p = xw[7] * y[5];
      acc[2] += 3.535533906e-01f * p;
    }
    { const T p = xw[7] * y[6];
      acc[1] += 2.886751346e-01f * p;
    }
    { const T p = xw[8] * y[0];
      acc[0] += -4.082482905e-01f * p;
    }
    { const T p = xw[8] * y[6];
      acc[2] += 4.082482905e-01f * p;
    }
}
template <class T> __device__ __forceinline__ void cg_432(const T* xw, const float* y, T* acc) {
    { const T p = xw[0] * y[0];
      acc[1] += 4.082482905e-01f * p;
    }
    { const T p = xw[0] * y[1];
      acc[0] += -3.333333333e-01f * p;
    }
    { const T p = xw[0] * y[5];
      acc[4] += 3.333333333e-01f * p;
    }
    { const T p = xw[0] * y[6];
      acc[3] += -4.082482905e-01f * p;
    }
    { const T p = xw[1] * y[1];
      acc[1] += 1.178511302e-01f * p;
    }
    { const T p = xw[1] * y[2];
      acc[0] += -3.726779962e-01f * p;
    }
    { const T p = xw[1] * y[4];
      acc[4] += 3.726779962e-01f * p;
    }
    { const T p = xw[1] * y[5];
      acc[3] += -1.178511302e-01f * p;
    }
    { const T p = xw[1] * y[6];
      acc[2] += -5.000000000e-01f * p;
    }
    { const T p = xw[2] * y[0];
      acc[1] += 2.314550249e-01f * p;
    }
    { const T p = xw[2] * y[2];
      acc[1] += -9.960238411e-02f * p;
    }
    { const T p = xw[2] * y[3];
      acc[4] += 4.879500365e-01f * p;
    }
    { const T p = xw[2] * y[4];
      acc[3] += 9.960238411e-02f * p;
    }
    { const T p = xw[2] * y[5];
      acc[2] += -4.364357805e-01f * p;
    }
    { const T p = xw[2] * y[6];
      acc[3] += 2.314550249e-01f * p;
    }
    { const T p = xw[3] * y[0];
      acc[0] += -1.091089451e-01f * p;
    }
    { const T p = xw[3] * y[1];
      acc[1] += 3.118047822e-01f * p;
    }
    { const T p = xw[3] * y[2];
      acc[0] += -2.817180849e-01f * p;
    }
    { const T p = xw[3] * y[3];
      acc[3] += 3.450327797e-01f * p;
    }
    { const T p = xw[3] * y[4];
      acc[2] += -2.439750182e-01f * p;
      acc[4] += -2.817180849e-01f * p;
    }
    { const T p = xw[3] * y[5];
      acc[3] += 3.118047822e-01f * p;
    }
    { const T p = xw[3] * y[6];
      acc[4] += -1.091089451e-01f * p;
    }
    { const T p = xw[4] * y[1];
      acc[4] += 2.817180849e-01f * p;
    }
    { const T p = xw[4] * y[2];
      acc[3] += -4.454354032e-01f * p;
    }
    { const T p = xw[4] * y[4];
      acc[1] += 4.454354032e-01f * p;
    }
    { const T p = xw[4] * y[5];
      acc[0] += -2.817180849e-01f * p;
    }
    { const T p = xw[5] * y[0];
      acc[4] += 1.091089451e-01f * p;
    }
    { const T p = xw[5] * y[1];
      acc[3] += -3.118047822e-01f * p;
    }
    { const T p = xw[5] * y[2];
      acc[2] += 2.439750182e-01f * p;
      acc[4] += -2.817180849e-01f * p;
    }
    { const T p = xw[5] * y[3];
      acc[1] += -3.450327797e-01f * p;
    }
    { const T p = xw[5] * y[4];
      acc[0] += 2.817180849e-01f * p;
    }
    { const T p = xw[5] * y[5];
      acc[1] += 3.118047822e-01f * p;
    }
    { const T p = xw[5] * y[6];
      acc[0] += -1.091089451e-01f * p;
    }
    { const T p = xw[6] * y[0];
      acc[3] += -2.314550249e-01f * p;
    }
    { const T p = xw[6] * y[1];
      acc[2] += 4.364357805e-01f * p;
    }
    { const T p = xw[6] * y[2];
      acc[3] += -9.960238411e-02f * p;
    }
    { const T p = xw[6] * y[3];
      acc[0] += -4.879500365e-01f * p;
    }
    { const T p = xw[6] * y[4];
      acc[1] += -9.960238411e-02f * p;
    }
    { const T p = xw[6] * y[6];
      acc[1] += 2.314550249e-01f * p;
    }
    { const T p = xw[7] * y[0];
      acc[2] += 5.000000000e-01f * p;
    }
    { const T p = xw[7] * y[1];
      acc[3] += 1.178511302e-01f * p;
    }
    { const T p = xw[7] * y[2];
      acc[4] += -3.726779962e-01f * p;
    }
    { const T p = xw[7] * y[4];
      acc[0] += -3.726779962e-01f * p;
    }
    { const T p = xw[7] * y[5];
      acc[1] += 1.178511302e-01f * p;
    }
    { const T p = xw[8] * y[0];
      acc[3] += 4.082482905e-01f * p;
    }
    { const T p = xw[8] * y[1];
      acc[4] += -3.333333333e-01f * p;
    }
    { const T p = xw[8] * y[5];
      acc[0] += -3.333333333e-01f * p;
    }
    { const T p = xw[8] * y[6];
      acc[1] += 4.082482905e-01f * p;
    }
}
template <class T> __device__ __forceinline__ void cg_433(const T* xw, const float* y, T* acc) {
    { const T p = xw[0] * y[0];
      acc[4] += -3.256694736e-01f * p;
    }
    { const T p = xw[0] * y[1];
      acc[5] += 4.204374826e-01f * p;
    }
    { const T p = xw[0] * y[2];
      acc[6] += -3.256694736e-01f * p;
    }
    { const T p = xw[0] * y[4];
      acc[0] += -3.256694736e-01f * p;
    }
    { const T p = xw[0] * y[5];
      acc[1] += 4.204374826e-01f * p;
    }
    { const T p = xw[0] * y[6];
      acc[2] += -3.256694736e-01f * p;
    }
    { const T p = xw[1] * y[0];
      acc[3] += -5.640760748e-01f * p;
    }
    { const T p = xw[1] * y[1];
      acc[4] += 1.880253583e-01f * p;
    }
    { const T p = xw[1] * y[2];
      acc[5] += 1.880253583e-01f * p;
    }
    { const T p = xw[1] * y[3];
      acc[0] += -5.640760748e-01f * p;
    }
    { const T p = xw[1] * y[4];
      acc[1] += 1.880253583e-01f * p;
    }
    { const T p = xw[1] * y[5];
      acc[2] += 1.880253583e-01f * p;
    }
    { const T p = xw[2] * y[0];
      acc[4] += 3.692744729e-01f * p;
    }
    { const T p = xw[2] * y[1];
      acc[3] += -1.230914910e-01f * p;
    }
    { const T p = xw[2] * y[2];
      acc[4] += 3.178208631e-01f * p;
      acc[6] += -3.692744729e-01f * p;
    }
    { const T p = xw[2] * y[3];
      acc[1] += -1.230914910e-01f * p;
    }
    { const T p = xw[2] * y[4];
      acc[0] += 3.692744729e-01f * p;
      acc[2] += 3.178208631e-01f * p;
    }
    { const T p = xw[2] * y[6];
      acc[2] += -3.692744729e-01f * p;
    }
    { const T p = xw[3] * y[0];
      acc[5] += -2.752409413e-01f * p;
    }
    { const T p = xw[3] * y[1];
      acc[4] += 2.842676218e-01f * p;
      acc[6] += 2.752409413e-01f * p;
    }
    { const T p = xw[3] * y[2];
      acc[3] += 2.752409413e-01f * p;
      acc[5] += -2.842676218e-01f * p;
    }
    { const T p = xw[3] * y[3];
      acc[2] += 2.752409413e-01f * p;
    }
    { const T p = xw[3] * y[4];
      acc[1] += 2.842676218e-01f * p;
    }
    { const T p = xw[3] * y[5];
      acc[0] += -2.752409413e-01f * p;
      acc[2] += -2.842676218e-01f * p;
    }
    { const T p = xw[3] * y[6];
      acc[1] += 2.752409413e-01f * p;
    }
    { const T p = xw[4] * y[0];
      acc[0] += 2.132007164e-01f * p;
    }
    { const T p = xw[4] * y[1];
      acc[1] += -4.974683382e-01f * p;
    }
    { const T p = xw[4] * y[2];
      acc[2] += 7.106690545e-02f * p;
    }
    { const T p = xw[4] * y[3];
      acc[3] += 4.264014327e-01f * p;
    }
    { const T p = xw[4] * y[4];
      acc[4] += 7.106690545e-02f * p;
    }
    { const T p = xw[4] * y[5];
      acc[5] += -4.974683382e-01f * p;
    }
    { const T p = xw[4] * y[6];
      acc[6] += 2.132007164e-01f * p;
    }
    { const T p = xw[5] * y[0];
      acc[1] += -2.752409413e-01f * p;
    }
    { const T p = xw[5] * y[1];
      acc[0] += -2.752409413e-01f * p;
      acc[2] += 2.842676218e-01f * p;
    }
    { const T p = xw[5] * y[2];
      acc[1] += 2.842676218e-01f * p;
    }
    { const T p = xw[5] * y[3];
      acc[4] += 2.752409413e-01f * p;
    }
    { const T p = xw[5] * y[4];
      acc[3] += 2.752409413e-01f * p;
      acc[5] += 2.842676218e-01f * p;
    }
    { const T p = xw[5] * y[5];
      acc[4] += 2.842676218e-01f * p;
      acc[6] += -2.752409413e-01f * p;
    }
    { const T p = xw[5] * y[6];
      acc[5] += -2.752409413e-01f * p;
    }
    { const T p = xw[6] * y[0];
      acc[2] += 3.692744729e-01f * p;
    }
    { const T p = xw[6] * y[2];
      acc[0] += 3.692744729e-01f * p;
      acc[2] += -3.178208631e-01f * p;
    }
    { const T p = xw[6] * y[3];
      acc[5] += -1.230914910e-01f * p;
    }
    { const T p = xw[6] * y[4];
      acc[4] += 3.178208631e-01f * p;
      acc[6] += 3.692744729e-01f * p;
    }
    { const T p = xw[6] * y[5];
      acc[3] += -1.230914910e-01f * p;
    }
    { const T p = xw[6] * y[6];
      acc[4] += 3.692744729e-01f * p;
    }
    { const T p = xw[7] * y[1];
      acc[2] += -1.880253583e-01f * p;
    }
    { const T p = xw[7] * y[2];
      acc[1] += -1.880253583e-01f * p;
    }
    { const T p = xw[7] * y[3];
      acc[6] += -5.640760748e-01f * p;
    }
    { const T p = xw[7] * y[4];
      acc[5] += 1.880253583e-01f * p;
    }
    { const T p = xw[7] * y[5];
      acc[4] += 1.880253583e-01f * p;
    }
    { const T p = xw[7] * y[6];
      acc[3] += -5.640760748e-01f * p;
    }
    { const T p = xw[8] * y[0];
      acc[2] += 3.256694736e-01f * p;
    }
    { const T p = xw[8] * y[1];
      acc[1] += -4.204374826e-01f * p;
    }
    { const T p = xw[8] * y[2];
      acc[0] += 3.256694736e-01f * p;
    }
    { const T p = xw[8] * y[4];
      acc[6] += -3.256694736e-01f * p;
    }
    { const T p = xw[8] * y[5];
      acc[5] += 4.204374826e-01f * p;
    }
    { const T p = xw[8] * y[6];
      acc[4] += -3.256694736e-01f * p;
    }
}
template <class T> __device__ __forceinline__ void cg_434(const T* xw, const float* y, T* acc) {
    { const T p = xw[0] * y[0];
      acc[3] += -2.132007164e-01f * p;
    }
    { const T p = xw[0] * y[1];
      acc[2] += 3.692744729e-01f * p;
    }
    { const T p = xw[0] * y[2];
      acc[1] += -4.369314488e-01f * p;
    }
    { const T p = xw[0] * y[3];
      acc[8] += -5.045249791e-01f * p;
    }
    { const T p = xw[0] * y[4];
      acc[7] += 4.369314488e-01f * p;
    }
    { const T p = xw[0] * y[5];
      acc[6] += -3.692744729e-01f * p;
    }
    { const T p = xw[0] * y[6];
      acc[5] += 2.132007164e-01f * p;
    }
    { const T p = xw[1] * y[1];
      acc[3] += 3.692744729e-01f * p;
    }
    { const T p = xw[1] * y[2];
      acc[0] += 4.369314488e-01f * p;
      acc[2] += -1.651445648e-01f * p;
    }
    { const T p = xw[1] * y[3];
      acc[7] += 2.522624896e-01f * p;
    }
    { const T p = xw[1] * y[4];
      acc[6] += 1.651445648e-01f * p;
      acc[8] += 4.369314488e-01f * p;
    }
    { const T p = xw[1] * y[5];
      acc[5] += -3.692744729e-01f * p;
    }
    { const T p = xw[1] * y[6];
      acc[4] += 4.767312946e-01f * p;
    }
    { const T p = xw[2] * y[0];
      acc[3] += -4.029114820e-01f * p;
    }
    { const T p = xw[2] * y[1];
      acc[0] += -3.692744729e-01f * p;
    }
    { const T p = xw[2] * y[2];
      acc[1] += 1.651445648e-01f * p;
      acc[3] += 1.248375568e-01f * p;
    }
    { const T p = xw[2] * y[3];
      acc[6] += 4.684874806e-01f * p;
    }
    { const T p = xw[2] * y[4];
      acc[5] += -1.248375568e-01f * p;
      acc[7] += 1.651445648e-01f * p;
    }
    { const T p = xw[2] * y[5];
      acc[4] += -3.120938920e-01f * p;
      acc[8] += -3.692744729e-01f * p;
    }
    { const T p = xw[2] * y[6];
      acc[5] += -4.029114820e-01f * p;
    }
    { const T p = xw[3] * y[0];
      acc[0] += 2.132007164e-01f * p;
      acc[2] += 4.029114820e-01f * p;
    }
    { const T p = xw[3] * y[1];
      acc[1] += -3.692744729e-01f * p;
    }
    { const T p = xw[3] * y[2];
      acc[2] += -1.248375568e-01f * p;
    }
    { const T p = xw[3] * y[3];
      acc[5] += 3.243374866e-01f * p;
    }
    { const T p = xw[3] * y[4];
      acc[4] += -4.187178947e-01f * p;
      acc[6] += -1.248375568e-01f * p;
    }
    { const T p = xw[3] * y[5];
      acc[7] += -3.692744729e-01f * p;
    }
    { const T p = xw[3] * y[6];
      acc[6] += 4.029114820e-01f * p;
      acc[8] += 2.132007164e-01f * p;
    }
    { const T p = xw[4] * y[0];
      acc[7] += 4.767312946e-01f * p;
    }
    { const T p = xw[4] * y[1];
      acc[6] += -3.120938920e-01f * p;
    }
    { const T p = xw[4] * y[2];
      acc[5] += -4.187178947e-01f * p;
    }
    { const T p = xw[4] * y[4];
      acc[3] += 4.187178947e-01f * p;
    }
    { const T p = xw[4] * y[5];
      acc[2] += 3.120938920e-01f * p;
    }
    { const T p = xw[4] * y[6];
      acc[1] += -4.767312946e-01f * p;
    }
    { const T p = xw[5] * y[0];
      acc[6] += -4.029114820e-01f * p;
      acc[8] += 2.132007164e-01f * p;
    }
    { const T p = xw[5] * y[1];
      acc[7] += -3.692744729e-01f * p;
    }
    { const T p = xw[5] * y[2];
      acc[4] += 4.187178947e-01f * p;
      acc[6] += -1.248375568e-01f * p;
    }
    { const T p = xw[5] * y[3];
      acc[3] += -3.243374866e-01f * p;
    }
    { const T p = xw[5] * y[4];
      acc[2] += 1.248375568e-01f * p;
    }
    { const T p = xw[5] * y[5];
      acc[1] += 3.692744729e-01f * p;
    }
    { const T p = xw[5] * y[6];
      acc[0] += -2.132007164e-01f * p;
      acc[2] += 4.029114820e-01f * p;
    }
    { const T p = xw[6] * y[0];
      acc[5] += 4.029114820e-01f * p;
    }
    { const T p = xw[6] * y[1];
      acc[4] += 3.120938920e-01f * p;
      acc[8] += -3.692744729e-01f * p;
    }
    { const T p = xw[6] * y[2];
      acc[5] += 1.248375568e-01f * p;
      acc[7] += 1.651445648e-01f * p;
    }
    { const T p = xw[6] * y[3];
      acc[2] += -4.684874806e-01f * p;
    }
    { const T p = xw[6] * y[4];
      acc[1] += -1.651445648e-01f * p;
      acc[3] += 1.248375568e-01f * p;
    }
    { const T p = xw[6] * y[5];
      acc[0] += 3.692744729e-01f * p;
    }
    { const T p = xw[6] * y[6];
      acc[3] += -4.029114820e-01f * p;
    }
    { const T p = xw[7] * y[0];
      acc[4] += -4.767312946e-01f * p;
    }
    { const T p = xw[7] * y[1];
      acc[5] += 3.692744729e-01f * p;
    }
    { const T p = xw[7] * y[2];
      acc[6] += -1.651445648e-01f * p;
      acc[8] += 4.369314488e-01f * p;
    }
    { const T p = xw[7] * y[3];
      acc[1] += -2.522624896e-01f * p;
    }
    { const T p = xw[7] * y[4];
      acc[0] += -4.369314488e-01f * p;
      acc[2] += -1.651445648e-01f * p;
    }
    { const T p = xw[7] * y[5];
      acc[3] += 3.692744729e-01f * p;
    }
    { const T p = xw[8] * y[0];
      acc[5] += -2.132007164e-01f * p;
    }
    { const T p = xw[8] * y[1];
      acc[6] += 3.692744729e-01f * p;
    }
    { const T p = xw[8] * y[2];
      acc[7] += -4.369314488e-01f * p;
    }
    { const T p = xw[8] * y[3];
      acc[0] += 5.045249791e-01f * p;
    }
    { const T p = xw[8] * y[4];
      acc[1] += -4.369314488e-01f * p;
    }
    { const T p = xw[8] * y[5];
      acc[2] += 3.692744729e-01f * p;
    }
    { const T p = xw[8] * y[6];
      acc[3] += -2.132007164e-01f * p;
    }
}
template <class T> __device__ __forceinline__ void cg_440(const T* xw, const float* y, T* acc) {
    { const T p = xw[0] * y[0];
      acc[0] += 3.333333333e-01f * p;
    }
    { const T p = xw[1] * y[1];
      acc[0] += 3.333333333e-01f * p;
    }
    { const T p = xw[2] * y[2];
      acc[0] += 3.333333333e-01f * p;
    }
    { const T p = xw[3] * y[3];
      acc[0] += 3.333333333e-01f * p;
    }
    { const T p = xw[4] * y[4];
      acc[0] += 3.333333333e-01f * p;
    }
    { const T p = xw[5] * y[5];
      acc[0] += 3.333333333e-01f * p;
    }
    { const T p = xw[6] * y[6];
      acc[0] += 3.333333333e-01f * p;
    }
    { const T p = xw[7] * y[7];
      acc[0] += 3.333333333e-01f * p;
    }
    { const T p = xw[8] * y[8];
      acc[0] += 3.333333333e-01f * p;
    }
}
template <class T> __device__ __forceinline__ void cg_441(const T* xw, const float* y, T* acc) {
    { const T p = xw[0] * y[1];
      acc[0] += -1.825741858e-01f * p;
    }
    { const T p = xw[0] * y[7];
      acc[2] += 1.825741858e-01f * p;
    }
    { const T p = xw[0] * y[8];
      acc[1] += -5.163977795e-01f * p;
    }
    { const T p = xw[1] * y[0];
      acc[0] += 1.825741858e-01f * p;
    }
    { const T p = xw[1] * y[2];
      acc[0] += -2.415229458e-01f * p;
    }
    { const T p = xw[1] * y[6];
      acc[2] += 2.415229458e-01f * p;
    }
    { const T p = xw[1] * y[7];
      acc[1] += -3.872983346e-01f * p;
    }
    { const T p = xw[1] * y[8];
      acc[2] += 1.825741858e-01f * p;
    }
    { const T p = xw[2] * y[1];
      acc[0] += 2.415229458e-01f * p;
    }
    { const T p = xw[2] * y[3];
      acc[0] += -2.738612788e-01f * p;
    }
    { const T p = xw[2] * y[5];
      acc[2] += 2.738612788e-01f * p;
    }
    { const T p = xw[2] * y[6];
      acc[1] += -2.581988897e-01f * p;
    }
    { const T p = xw[2] * y[7];
      acc[2] += 2.415229458e-01f * p;
    }
    { const T p = xw[3] * y[2];
      acc[0] += 2.738612788e-01f * p;
    }
    { const T p = xw[3] * y[4];
      acc[2] += 4.082482905e-01f * p;
    }
    { const T p = xw[3] * y[5];
      acc[1] += -1.290994449e-01f * p;
    }
    { const T p = xw[3] * y[6];
      acc[2] += 2.738612788e-01f * p;
    }
    { const T p = xw[4] * y[3];
      acc[2] += -4.082482905e-01f * p;
    }
    { const T p = xw[4] * y[5];
      acc[0] += 4.082482905e-01f * p;
    }
    { const T p = xw[5] * y[2];
      acc[2] += -2.738612788e-01f * p;
    }
    { const T p = xw[5] * y[3];
      acc[1] += 1.290994449e-01f * p;
    }
    { const T p = xw[5] * y[4];
      acc[0] += -4.082482905e-01f * p;
    }
    { const T p = xw[5] * y[6];
      acc[0] += 2.738612788e-01f * p;
    }
    { const T p = xw[6] * y[1];
      acc[2] += -2.415229458e-01f * p;
    }
    { const T p = xw[6] * y[2];
      acc[1] += 2.581988897e-01f * p;
    }
    { const T p = xw[6] * y[3];
      acc[2] += -2.738612788e-01f * p;
    }
    { const T p = xw[6] * y[5];
      acc[0] += -2.738612788e-01f * p;
    }
    { const T p = xw[6] * y[7];
      acc[0] += 2.415229458e-01f * p;
    }
    { const T p = xw[7] * y[0];
      acc[2] += -1.825741858e-01f * p;
    }
    { const T p = xw[7] * y[1];
      acc[1] += 3.872983346e-01f * p;
    }
    { const T p = xw[7] * y[2];
      acc[2] += -2.415229458e-01f * p;
    }
    { const T p = xw[7] * y[6];
      acc[0] += -2.415229458e-01f * p;
    }
    { const T p = xw[7] * y[8];
      acc[0] += 1.825741858e-01f * p;
    }
    { const T p = xw[8] * y[0];
      acc[1] += 5.163977795e-01f * p;
    }
    { const T p = xw[8] * y[1];
      acc[2] += -1.825741858e-01f * p;
    }
    { const T p = xw[8] * y[7];
      acc[0] += -1.825741858e-01f * p;
    }
}
template <class T> __device__ __forceinline__ void cg_442(const T* xw, const float* y, T* acc) {
    { const T p = xw[0] * y[0];
      acc[2] += -5.318160235e-01f * p;
    }
    { const T p = xw[0] * y[1];
      acc[3] += 3.256694736e-01f * p;
    }
    { const T p = xw[0] * y[2];
      acc[4] += -1.740776560e-01f * p;
    }
    { const T p = xw[0] * y[6];
      acc[0] += -1.740776560e-01f * p;
    }
    { const T p = xw[0] * y[7];
      acc[1] += 3.256694736e-01f * p;
    }
    { const T p = xw[1] * y[0];
      acc[3] += 3.256694736e-01f * p;
    }
    { const T p = xw[1] * y[1];
      acc[2] += -1.329540059e-01f * p;
    }
    { const T p = xw[1] * y[2];
      acc[3] += 3.077287274e-01f * p;
    }
    { const T p = xw[1] * y[3];
      acc[4] += -2.611164839e-01f * p;
    }
    { const T p = xw[1] * y[5];
      acc[0] += -2.611164839e-01f * p;
    }
    { const T p = xw[1] * y[6];
      acc[1] += 3.077287274e-01f * p;
    }
    { const T p = xw[1] * y[8];
      acc[1] += -3.256694736e-01f * p;
    }
    { const T p = xw[2] * y[0];
      acc[4] += -1.740776560e-01f * p;
    }
    { const T p = xw[2] * y[1];
      acc[3] += 3.077287274e-01f * p;
    }
    { const T p = xw[2] * y[2];
      acc[2] += 1.519474353e-01f * p;
    }
    { const T p = xw[2] * y[3];
      acc[3] += 2.093589473e-01f * p;
    }
    { const T p = xw[2] * y[4];
      acc[0] += -4.413674148e-01f * p;
    }
    { const T p = xw[2] * y[5];
      acc[1] += 2.093589473e-01f * p;
    }
    { const T p = xw[2] * y[7];
      acc[1] += -3.077287274e-01f * p;
    }
    { const T p = xw[2] * y[8];
      acc[0] += 1.740776560e-01f * p;
    }
    { const T p = xw[3] * y[1];
      acc[4] += -2.611164839e-01f * p;
    }
    { const T p = xw[3] * y[2];
      acc[3] += 2.093589473e-01f * p;
    }
    { const T p = xw[3] * y[3];
      acc[2] += 3.228883000e-01f * p;
      acc[4] += -3.289758475e-01f * p;
    }
    { const T p = xw[3] * y[4];
      acc[1] += 1.040312973e-01f * p;
    }
    { const T p = xw[3] * y[5];
      acc[0] += 3.289758475e-01f * p;
    }
    { const T p = xw[3] * y[6];
      acc[1] += -2.093589473e-01f * p;
    }
    { const T p = xw[3] * y[7];
      acc[0] += 2.611164839e-01f * p;
    }
    { const T p = xw[4] * y[2];
      acc[0] += -4.413674148e-01f * p;
    }
    { const T p = xw[4] * y[3];
      acc[1] += 1.040312973e-01f * p;
    }
    { const T p = xw[4] * y[4];
      acc[2] += 3.798685882e-01f * p;
    }
    { const T p = xw[4] * y[5];
      acc[3] += 1.040312973e-01f * p;
    }
    { const T p = xw[4] * y[6];
      acc[4] += -4.413674148e-01f * p;
    }
    { const T p = xw[5] * y[1];
      acc[0] += -2.611164839e-01f * p;
    }
    { const T p = xw[5] * y[2];
      acc[1] += 2.093589473e-01f * p;
    }
    { const T p = xw[5] * y[3];
      acc[0] += 3.289758475e-01f * p;
    }
    { const T p = xw[5] * y[4];
      acc[3] += 1.040312973e-01f * p;
    }
    { const T p = xw[5] * y[5];
      acc[2] += 3.228883000e-01f * p;
      acc[4] += 3.289758475e-01f * p;
    }
    { const T p = xw[5] * y[6];
      acc[3] += 2.093589473e-01f * p;
    }
    { const T p = xw[5] * y[7];
      acc[4] += -2.611164839e-01f * p;
    }
    { const T p = xw[6] * y[0];
      acc[0] += -1.740776560e-01f * p;
    }
    { const T p = xw[6] * y[1];
      acc[1] += 3.077287274e-01f * p;
    }
    { const T p = xw[6] * y[3];
      acc[1] += -2.093589473e-01f * p;
    }
    { const T p = xw[6] * y[4];
      acc[4] += -4.413674148e-01f * p;
    }
    { const T p = xw[6] * y[5];
      acc[3] += 2.093589473e-01f * p;
    }
    { const T p = xw[6] * y[6];
      acc[2] += 1.519474353e-01f * p;
    }
    { const T p = xw[6] * y[7];
      acc[3] += 3.077287274e-01f * p;
    }
    { const T p = xw[6] * y[8];
      acc[4] += -1.740776560e-01f * p;
    }
    { const T p = xw[7] * y[0];
      acc[1] += 3.256694736e-01f * p;
    }
    { const T p = xw[7] * y[2];
      acc[1] += -3.077287274e-01f * p;
    }
    { const T p = xw[7] * y[3];
      acc[0] += 2.611164839e-01f * p;
    }
    { const T p = xw[7] * y[5];
      acc[4] += -2.611164839e-01f * p;
    }
    { const T p = xw[7] * y[6];
      acc[3] += 3.077287274e-01f * p;
    }
    { const T p = xw[7] * y[7];
      acc[2] += -1.329540059e-01f * p;
    }
    { const T p = xw[7] * y[8];
      acc[3] += 3.256694736e-01f * p;
    }
    { const T p = xw[8] * y[1];
      acc[1] += -3.256694736e-01f * p;
    }
    { const T p = xw[8] * y[2];
      acc[0] += 1.740776560e-01f * p;
    }
    { const T p = xw[8] * y[6];
      acc[4] += -1.740776560e-01f * p;
    }
    { const T p = xw[8] * y[7];
      acc[3] += 3.256694736e-01f * p;
    }
    { const T p = xw[8] * y[8];
      acc[2] += -5.318160235e-01f * p;
    }
}
template <class T> __device__ __forceinline__ void cg_443(const T* xw, const float* y, T* acc) {
    { const T p = xw[0] * y[1];
      acc[2] += 3.853373178e-01f * p;
    }
    { const T p = xw[0] * y[2];
      acc[1] += -3.256694736e-01f * p;
    }
    { const T p = xw[0] * y[3];
      acc[0] += 1.880253583e-01f * p;
    }
    { const T p = xw[0] * y[5];
      acc[6] += -1.880253583e-01f * p;
    }
    { const T p = xw[0] * y[6];
      acc[5] += 3.256694736e-01f * p;
    }
    { const T p = xw[0] * y[7];
      acc[4] += -3.853373178e-01f * p;
    }
    { const T p = xw[0] * y[8];
      acc[3] += 4.449492083e-01f * p;
    }
    { const T p = xw[1] * y[0];
      acc[2] += -3.853373178e-01f * p;
    }
    { const T p = xw[1] * y[2];
      acc[2] += 1.456438163e-01f * p;
    }
    { const T p = xw[1] * y[3];
      acc[1] += -3.256694736e-01f * p;
    }
    { const T p = xw[1] * y[4];
      acc[6] += -4.204374826e-01f * p;
    }
    { const T p = xw[1] * y[5];
      acc[5] += 3.256694736e-01f * p;
    }
    { const T p = xw[1] * y[6];
      acc[4] += -1.456438163e-01f * p;
    }
    { const T p = xw[1] * y[7];
      acc[3] += -2.224746042e-01f * p;
    }
    { const T p = xw[1] * y[8];
      acc[4] += -3.853373178e-01f * p;
    }
    { const T p = xw[2] * y[0];
      acc[1] += 3.256694736e-01f * p;
    }
    { const T p = xw[2] * y[1];
      acc[2] += -1.456438163e-01f * p;
    }
    { const T p = xw[2] * y[3];
      acc[0] += 3.553345273e-01f * p;
      acc[2] += -1.100963765e-01f * p;
    }
    { const T p = xw[2] * y[4];
      acc[5] += 2.752409413e-01f * p;
    }
    { const T p = xw[2] * y[5];
      acc[4] += 1.100963765e-01f * p;
      acc[6] += 3.553345273e-01f * p;
    }
    { const T p = xw[2] * y[6];
      acc[3] += -4.131671220e-01f * p;
    }
    { const T p = xw[2] * y[7];
      acc[4] += -1.456438163e-01f * p;
    }
    { const T p = xw[2] * y[8];
      acc[5] += 3.256694736e-01f * p;
    }
    { const T p = xw[3] * y[0];
      acc[0] += -1.880253583e-01f * p;
    }
    { const T p = xw[3] * y[1];
      acc[1] += 3.256694736e-01f * p;
    }
    { const T p = xw[3] * y[2];
      acc[0] += -3.553345273e-01f * p;
      acc[2] += 1.100963765e-01f * p;
    }
    { const T p = xw[3] * y[4];
      acc[4] += 3.692744729e-01f * p;
    }
    { const T p = xw[3] * y[5];
      acc[3] += -2.860387768e-01f * p;
    }
    { const T p = xw[3] * y[6];
      acc[4] += 1.100963765e-01f * p;
      acc[6] += -3.553345273e-01f * p;
    }
    { const T p = xw[3] * y[7];
      acc[5] += 3.256694736e-01f * p;
    }
    { const T p = xw[3] * y[8];
      acc[6] += -1.880253583e-01f * p;
    }
    { const T p = xw[4] * y[1];
      acc[6] += 4.204374826e-01f * p;
    }
    { const T p = xw[4] * y[2];
      acc[5] += -2.752409413e-01f * p;
    }
    { const T p = xw[4] * y[3];
      acc[4] += -3.692744729e-01f * p;
    }
    { const T p = xw[4] * y[5];
      acc[2] += 3.692744729e-01f * p;
    }
    { const T p = xw[4] * y[6];
      acc[1] += 2.752409413e-01f * p;
    }
    { const T p = xw[4] * y[7];
      acc[0] += -4.204374826e-01f * p;
    }
    { const T p = xw[5] * y[0];
      acc[6] += 1.880253583e-01f * p;
    }
    { const T p = xw[5] * y[1];
      acc[5] += -3.256694736e-01f * p;
    }
    { const T p = xw[5] * y[2];
      acc[4] += -1.100963765e-01f * p;
      acc[6] += -3.553345273e-01f * p;
    }
    { const T p = xw[5] * y[3];
      acc[3] += 2.860387768e-01f * p;
    }
    { const T p = xw[5] * y[4];
      acc[2] += -3.692744729e-01f * p;
    }
    { const T p = xw[5] * y[6];
      acc[0] += 3.553345273e-01f * p;
      acc[2] += 1.100963765e-01f * p;
    }
    { const T p = xw[5] * y[7];
      acc[1] += 3.256694736e-01f * p;
    }
    { const T p = xw[5] * y[8];
      acc[0] += -1.880253583e-01f * p;
    }
    { const T p = xw[6] * y[0];
      acc[5] += -3.256694736e-01f * p;
    }
    { const T p = xw[6] * y[1];
      acc[4] += 1.456438163e-01f * p;
    }
    { const T p = xw[6] * y[2];
      acc[3] += 4.131671220e-01f * p;
    }
    { const T p = xw[6] * y[3];
      acc[4] += -1.100963765e-01f * p;
      acc[6] += 3.553345273e-01f * p;
    }
    { const T p = xw[6] * y[4];
      acc[1] += -2.752409413e-01f * p;
    }
    { const T p = xw[6] * y[5];
      acc[0] += -3.553345273e-01f * p;
      acc[2] += -1.100963765e-01f * p;
    }
    { const T p = xw[6] * y[7];
      acc[2] += -1.456438163e-01f * p;
    }
    { const T p = xw[6] * y[8];
      acc[1] += 3.256694736e-01f * p;
    }
    { const T p = xw[7] * y[0];
      acc[4] += 3.853373178e-01f * p;
    }
    { const T p = xw[7] * y[1];
      acc[3] += 2.224746042e-01f * p;
    }
    { const T p = xw[7] * y[2];
      acc[4] += 1.456438163e-01f * p;
    }
    { const T p = xw[7] * y[3];
      acc[5] += -3.256694736e-01f * p;
    }
    { const T p = xw[7] * y[4];
      acc[0] += 4.204374826e-01f * p;
    }
    { const T p = xw[7] * y[5];
      acc[1] += -3.256694736e-01f * p;
    }
    { const T p = xw[7] * y[6];
      acc[2] += 1.456438163e-01f * p;
    }
    { const T p = xw[7] * y[8];
      acc[2] += -3.853373178e-01f * p;
    }
    { const T p = xw[8] * y[0];
      acc[3] += -4.449492083e-01f * p;
    }
    { const T p = xw[8] * y[1];
      acc[4] += 3.853373178e-01f * p;
    }
    { const T p = xw[8] * y[2];
      acc[5] += -3.256694736e-01f * p;
    }
    { const T p = xw[8] * y[3];
      acc[6] += 1.880253583e-01f * p;
    }
    { const T p = xw[8] * y[5];
      acc[0] += 1.880253583e-01f * p;
    }
    { const T p = xw[8] * y[6];
      acc[1] += -3.256694736e-01f * p;
    }
    { const T p = xw[8] * y[7];
      acc[2] += 3.853373178e-01f * p;
    }
}
template <class T> __device__ __forceinline__ void cg_444(const T* xw, const float* y, T* acc) {
    { const T p = xw[0] * y[0];
      acc[4] += 3.128931094e-01f * p;
    }
    { const T p = xw[0] * y[1];
      acc[5] += -3.498251311e-01f * p;
    }
    { const T p = xw[0] * y[2];
      acc[6] += 3.966644140e-01f * p;
    }
    { const T p = xw[0] * y[3];
      acc[7] += -3.498251311e-01f * p;
    }
    { const T p = xw[0] * y[4];
      acc[0] += 3.128931094e-01f * p;
    }
    { const T p = xw[0] * y[5];
      acc[1] += -3.498251311e-01f * p;
    }
    { const T p = xw[0] * y[6];
      acc[2] += 3.966644140e-01f * p;
    }
    { const T p = xw[0] * y[7];
      acc[3] += -3.498251311e-01f * p;
    }
    { const T p = xw[1] * y[0];
      acc[5] += -3.498251311e-01f * p;
    }
    { const T p = xw[1] * y[1];
      acc[4] += -4.693396641e-01f * p;
    }
    { const T p = xw[1] * y[2];
      acc[5] += 1.322214713e-01f * p;
    }
    { const T p = xw[1] * y[3];
      acc[6] += 1.322214713e-01f * p;
      acc[8] += 3.498251311e-01f * p;
    }
    { const T p = xw[1] * y[4];
      acc[1] += -4.693396641e-01f * p;
    }
    { const T p = xw[1] * y[5];
      acc[0] += -3.498251311e-01f * p;
      acc[2] += 1.322214713e-01f * p;
    }
    { const T p = xw[1] * y[6];
      acc[3] += 1.322214713e-01f * p;
    }
    { const T p = xw[1] * y[8];
      acc[3] += 3.498251311e-01f * p;
    }
    { const T p = xw[2] * y[0];
      acc[6] += 3.966644140e-01f * p;
    }
    { const T p = xw[2] * y[1];
      acc[5] += 1.322214713e-01f * p;
    }
    { const T p = xw[2] * y[2];
      acc[4] += -2.458445859e-01f * p;
      acc[8] += -3.966644140e-01f * p;
    }
    { const T p = xw[2] * y[3];
      acc[5] += 2.998501124e-01f * p;
      acc[7] += -1.322214713e-01f * p;
    }
    { const T p = xw[2] * y[4];
      acc[2] += -2.458445859e-01f * p;
    }
    { const T p = xw[2] * y[5];
      acc[1] += 1.322214713e-01f * p;
      acc[3] += 2.998501124e-01f * p;
    }
    { const T p = xw[2] * y[6];
      acc[0] += 3.966644140e-01f * p;
    }
    { const T p = xw[2] * y[7];
      acc[3] += -1.322214713e-01f * p;
    }
    { const T p = xw[2] * y[8];
      acc[2] += -3.966644140e-01f * p;
    }
    { const T p = xw[3] * y[0];
      acc[7] += -3.498251311e-01f * p;
    }
    { const T p = xw[3] * y[1];
      acc[6] += 1.322214713e-01f * p;
      acc[8] += 3.498251311e-01f * p;
    }
    { const T p = xw[3] * y[2];
      acc[5] += 2.998501124e-01f * p;
      acc[7] += -1.322214713e-01f * p;
    }
    { const T p = xw[3] * y[3];
      acc[4] += 2.011455703e-01f * p;
      acc[6] += -2.998501124e-01f * p;
    }
    { const T p = xw[3] * y[4];
      acc[3] += 2.011455703e-01f * p;
    }
    { const T p = xw[3] * y[5];
      acc[2] += 2.998501124e-01f * p;
    }
    { const T p = xw[3] * y[6];
      acc[1] += 1.322214713e-01f * p;
      acc[3] += -2.998501124e-01f * p;
    }
    { const T p = xw[3] * y[7];
      acc[0] += -3.498251311e-01f * p;
      acc[2] += -1.322214713e-01f * p;
    }
    { const T p = xw[3] * y[8];
      acc[1] += 3.498251311e-01f * p;
    }
    { const T p = xw[4] * y[0];
      acc[0] += 3.128931094e-01f * p;
    }
    { const T p = xw[4] * y[1];
      acc[1] += -4.693396641e-01f * p;
    }
    { const T p = xw[4] * y[2];
      acc[2] += -2.458445859e-01f * p;
    }
    { const T p = xw[4] * y[3];
      acc[3] += 2.011455703e-01f * p;
    }
    { const T p = xw[4] * y[4];
      acc[4] += 4.022911406e-01f * p;
    }
    { const T p = xw[4] * y[5];
      acc[5] += 2.011455703e-01f * p;
    }
    { const T p = xw[4] * y[6];
      acc[6] += -2.458445859e-01f * p;
    }
    { const T p = xw[4] * y[7];
      acc[7] += -4.693396641e-01f * p;
    }
    { const T p = xw[4] * y[8];
      acc[8] += 3.128931094e-01f * p;
    }
    { const T p = xw[5] * y[0];
      acc[1] += -3.498251311e-01f * p;
    }
    { const T p = xw[5] * y[1];
      acc[0] += -3.498251311e-01f * p;
      acc[2] += 1.322214713e-01f * p;
    }
    { const T p = xw[5] * y[2];
      acc[1] += 1.322214713e-01f * p;
      acc[3] += 2.998501124e-01f * p;
    }
    { const T p = xw[5] * y[3];
      acc[2] += 2.998501124e-01f * p;
    }
    { const T p = xw[5] * y[4];
      acc[5] += 2.011455703e-01f * p;
    }
    { const T p = xw[5] * y[5];
      acc[4] += 2.011455703e-01f * p;
      acc[6] += 2.998501124e-01f * p;
    }
    { const T p = xw[5] * y[6];
      acc[5] += 2.998501124e-01f * p;
      acc[7] += 1.322214713e-01f * p;
    }
    { const T p = xw[5] * y[7];
      acc[6] += 1.322214713e-01f * p;
      acc[8] += -3.498251311e-01f * p;
    }
    { const T p = xw[5] * y[8];
      acc[7] += -3.498251311e-01f * p;
    }
    { const T p = xw[6] * y[0];
      acc[2] += 3.966644140e-01f * p;
    }
    { const T p = xw[6] * y[1];
      acc[3] += 1.322214713e-01f * p;
    }
    { const T p = xw[6] * y[2];
      acc[0] += 3.966644140e-01f * p;
    }
    { const T p = xw[6] * y[3];
      acc[1] += 1.322214713e-01f * p;
      acc[3] += -2.998501124e-01f * p;
    }
    { const T p = xw[6] * y[4];
      acc[6] += -2.458445859e-01f * p;
    }
    { const T p = xw[6] * y[5];
      acc[5] += 2.998501124e-01f * p;
      acc[7] += 1.322214713e-01f * p;
    }
    { const T p = xw[6] * y[6];
      acc[4] += -2.458445859e-01f * p;
      acc[8] += 3.966644140e-01f * p;
    }
    { const T p = xw[6] * y[7];
      acc[5] += 1.322214713e-01f * p;
    }
    { const T p = xw[6] * y[8];
      acc[6] += 3.966644140e-01f * p;
    }
    { const T p = xw[7] * y[0];
      acc[3] += -3.498251311e-01f * p;
    }
    { const T p = xw[7] * y[2];
      acc[3] += -1.322214713e-01f * p;
    }
    { const T p = xw[7] * y[3];
      acc[0] += -3.498251311e-01f * p;
      acc[2] += -1.322214713e-01f * p;
    }
    { const T p = xw[7] * y[4];
      acc[7] += -4.693396641e-01f * p;
    }
    { const T p = xw[7] * y[5];
      acc[6] += 1.322214713e-01f * p;
      acc[8] += -3.498251311e-01f * p;
    }
    { const T p = xw[7] * y[6];
      acc[5] += 1.322214713e-01f * p;
    }
    { const T p = xw[7] * y[7];
      acc[4] += -4.693396641e-01f * p;
    }
    { const T p = xw[7] * y[8];
      acc[5] += -3.498251311e-01f * p;
    }
    { const T p = xw[8] * y[1];
      acc[3] += 3.498251311e-01f * p;
    }
    { const T p = xw[8] * y[2];
      acc[2] += -3.966644140e-01f * p;
    }
    { const T p = xw[8] * y[3];
      acc[1] += 3.498251311e-01f * p;
    }
    { const T p = xw[8] * y[4];
      acc[8] += 3.128931094e-01f * p;
    }
    { const T p = xw[8] * y[5];
      acc[7] += -3.498251311e-01f * p;
    }
    { const T p = xw[8] * y[6];
      acc[6] += 3.966644140e-01f * p;
    }
    { const T p = xw[8] * y[7];
      acc[5] += -3.498251311e-01f * p;
    }
    { const T p = xw[8] * y[8];
      acc[4] += 3.128931094e-01f * p;
    }
}
constexpr int NC=9, NACC=49, D1=9, NY=16, OPS=1029;
template <class T> __device__ __forceinline__ void apply(const T* x, const float* y, const T* w, T* acc) {
    { T xw[9]; for (int i = 0; i < 9; ++i) xw[i] = w[0] * x[i]; cg_431<T>(xw, y + 0, acc + 0); }
    { T xw[9]; for (int i = 0; i < 9; ++i) xw[i] = w[1] * x[i]; cg_432<T>(xw, y + 0, acc + 3); }
    { T xw[9]; for (int i = 0; i < 9; ++i) xw[i] = w[2] * x[i]; cg_433<T>(xw, y + 0, acc + 8); }
    { T xw[9]; for (int i = 0; i < 9; ++i) xw[i] = w[3] * x[i]; cg_434<T>(xw, y + 0, acc + 15); }
    { T xw[9]; for (int i = 0; i < 9; ++i) xw[i] = w[4] * x[i]; cg_440<T>(xw, y + 7, acc + 24); }
    { T xw[9]; for (int i = 0; i < 9; ++i) xw[i] = w[5] * x[i]; cg_441<T>(xw, y + 7, acc + 25); }
    { T xw[9]; for (int i = 0; i < 9; ++i) xw[i] = w[6] * x[i]; cg_442<T>(xw, y + 7, acc + 28); }
    { T xw[9]; for (int i = 0; i < 9; ++i) xw[i] = w[7] * x[i]; cg_443<T>(xw, y + 7, acc + 33); }
    { T xw[9]; for (int i = 0; i < 9; ++i) xw[i] = w[8] * x[i]; cg_444<T>(xw, y + 7, acc + 40); }
}

template <class T, int MINB> __global__ __launch_bounds__(256, MINB) void k(const float* in, float* out, int iters) {
    const int l = blockIdx.x * 256 + threadIdx.x;
    T x[D1], w[NC], acc[NACC]; float y[NY];
    for (int i = 0; i < D1; ++i) x[i] = T(in[(l + i) & 1023]);
    for (int i = 0; i < NC; ++i) w[i] = T(in[(l + 7 * i) & 1023]);
    for (int i = 0; i < NY; ++i) y[i] = in[(3 * i + (l >> 6)) & 1023];
    for (int i = 0; i < NACC; ++i) acc[i] = T(0.f);
    for (int it = 0; it < iters; ++it) {
        apply<T>(x, y, w, acc);
        for (int i = 0; i < D1; ++i) x[i] += T(1e-3f);   // new "edge"
        y[it % NY] += 1e-3f;
    }
    T s = T(0.f);
    for (int i = 0; i < NACC; ++i) s += acc[i];
    out[l] = sizeof(T) == 4 ? *(float*)&s : ((float*)&s)[0] + ((float*)&s)[1];
}
int main() {
    float *in, *out; hipMalloc(&in, 4096); hipMalloc(&out, 1 << 22);
    float h[1024]; for (int i = 0; i < 1024; ++i) h[i] = 0.001f * (i % 97) - 0.03f;
    hipMemcpy(in, h, 4096, hipMemcpyHostToDevice);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    const int iters = 2000;
    auto run = [&](auto kern, int blocks, int chan_per_lane, const char* name) {
        kern<<<blocks, 256>>>(in, out, 10); hipDeviceSynchronize();
        hipEventRecord(a); kern<<<blocks, 256>>>(in, out, iters); hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        double chan_edges = (double)blocks * 256 * chan_per_lane * iters;
        printf("%-28s %8.3f ms  %.2f G channel-edges/s  (%.1f TFLOP/s of the %d-op schedule)\n", name, ms, chan_edges / ms / 1e6,
               chan_edges * OPS * 2 / ms / 1e9, OPS);
    };
    const int full = 256 * 4 * 3;   // 3 waves per SIMD on 256 CUs (blocks of 4 waves)
    run(k<float, 3>, full, 1, "scalar, 3 waves/SIMD");
    run(k<f2, 3>, full / 2, 2, "packed x2, 1.5 waves/SIMD eq.");
    run(k<f2, 3>, full, 2, "packed x2, 3 waves/SIMD");
    run(k<f2, 2>, 256 * 4 * 2, 2, "packed x2, 2 waves/SIMD");
    run(k<float, 4>, 256 * 4 * 4, 1, "scalar, 4 waves/SIMD");
    return 0;
}
