/* oracle/shim/cudnn.h — TEST INFRASTRUCTURE ONLY.
 *
 * Just enough of the cuDNN C interface (types, enumerators, descriptor life-cycle calls) for the reference's
 * src/cudnn_traits.hpp and src/cudnn_kernel_pool.hpp to compile verbatim: peak_finder_t owns a
 * Pool_NCHW_PaddingSame_Max member (src/post_process.hpp:215) whose constructor creates a handle and three
 * descriptors.  The pooling itself (cudnnPoolingForward) is only reached from the `use_gpu` branch, which
 * paf::process never takes (src/paf.cpp:345: `false`); it aborts here. */
#pragma once
#include <cstdio>
#include <cstdlib>

typedef enum { CUDNN_STATUS_SUCCESS = 0, CUDNN_STATUS_NOT_SUPPORTED = 9 } cudnnStatus_t;
typedef enum { CUDNN_TENSOR_NCHW = 0, CUDNN_TENSOR_NHWC = 1 } cudnnTensorFormat_t;
typedef enum { CUDNN_DATA_FLOAT = 0, CUDNN_DATA_DOUBLE = 1 } cudnnDataType_t;
typedef enum { CUDNN_CONVOLUTION = 0, CUDNN_CROSS_CORRELATION = 1 } cudnnConvolutionMode_t;
typedef enum { CUDNN_POOLING_MAX = 0 } cudnnPoolingMode_t;
typedef enum { CUDNN_NOT_PROPAGATE_NAN = 0, CUDNN_PROPAGATE_NAN = 1 } cudnnNanPropagation_t;

struct cudnnContext {
    int unused;
};
struct cudnnTensorStruct {
    cudnnDataType_t type;
    int n, c, h, w;
};
struct cudnnFilterStruct {
    int unused;
};
struct cudnnConvolutionStruct {
    int unused;
};
struct cudnnPoolingStruct {
    int unused;
};
typedef cudnnContext* cudnnHandle_t;
typedef cudnnTensorStruct* cudnnTensorDescriptor_t;
typedef cudnnFilterStruct* cudnnFilterDescriptor_t;
typedef cudnnConvolutionStruct* cudnnConvolutionDescriptor_t;
typedef cudnnPoolingStruct* cudnnPoolingDescriptor_t;

inline cudnnStatus_t cudnnCreate(cudnnHandle_t* h) { return *h = new cudnnContext(), CUDNN_STATUS_SUCCESS; }
inline cudnnStatus_t cudnnDestroy(cudnnHandle_t h) { return delete h, CUDNN_STATUS_SUCCESS; }
inline cudnnStatus_t cudnnCreateTensorDescriptor(cudnnTensorDescriptor_t* d) { return *d = new cudnnTensorStruct(), CUDNN_STATUS_SUCCESS; }
inline cudnnStatus_t cudnnDestroyTensorDescriptor(cudnnTensorDescriptor_t d) { return delete d, CUDNN_STATUS_SUCCESS; }
inline cudnnStatus_t cudnnDestroyFilterDescriptor(cudnnFilterDescriptor_t d) { return delete d, CUDNN_STATUS_SUCCESS; }
inline cudnnStatus_t cudnnDestroyConvolutionDescriptor(cudnnConvolutionDescriptor_t d) { return delete d, CUDNN_STATUS_SUCCESS; }
inline cudnnStatus_t cudnnCreatePoolingDescriptor(cudnnPoolingDescriptor_t* d) { return *d = new cudnnPoolingStruct(), CUDNN_STATUS_SUCCESS; }
inline cudnnStatus_t cudnnDestroyPoolingDescriptor(cudnnPoolingDescriptor_t d) { return delete d, CUDNN_STATUS_SUCCESS; }
inline cudnnStatus_t cudnnSetTensor4dDescriptor(cudnnTensorDescriptor_t d, cudnnTensorFormat_t, cudnnDataType_t t, int n, int c, int h, int w)
{
    d->type = t, d->n = n, d->c = c, d->h = h, d->w = w;
    return CUDNN_STATUS_SUCCESS;
}
inline cudnnStatus_t cudnnGetTensor4dDescriptor(const cudnnTensorDescriptor_t d, cudnnDataType_t* t, int* n, int* c, int* h, int* w, int* ns, int* cs, int* hs, int* ws)
{
    *t = d->type, *n = d->n, *c = d->c, *h = d->h, *w = d->w;
    *ws = 1, *hs = d->w, *cs = d->h * d->w, *ns = d->c * d->h * d->w;
    return CUDNN_STATUS_SUCCESS;
}
inline cudnnStatus_t cudnnSetPoolingNdDescriptor(cudnnPoolingDescriptor_t, cudnnPoolingMode_t, cudnnNanPropagation_t, int, const int*, const int*, const int*) { return CUDNN_STATUS_SUCCESS; }
/* 3x3, stride 1, pad 1: output dims = input dims */
inline cudnnStatus_t cudnnGetPooling2dForwardOutputDim(const cudnnPoolingDescriptor_t, const cudnnTensorDescriptor_t x, int* n, int* c, int* h, int* w)
{
    *n = x->n, *c = x->c, *h = x->h, *w = x->w;
    return CUDNN_STATUS_SUCCESS;
}
inline cudnnStatus_t cudnnPoolingForward(cudnnHandle_t, const cudnnPoolingDescriptor_t, const void*, const cudnnTensorDescriptor_t, const void*, const void*, const cudnnTensorDescriptor_t, void*)
{
    std::fprintf(stderr, "oracle/shim/cudnn.h: cudnnPoolingForward reached (the reference's use_gpu branch): not provided\n");
    std::abort();
}
