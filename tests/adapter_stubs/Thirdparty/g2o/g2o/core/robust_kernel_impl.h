// compile-hygiene stand-in (tests/adapter_stubs/README.md): core/robust_kernel.h:72, core/robust_kernel_impl.h
#pragma once
namespace g2o {
class RobustKernel { public: virtual ~RobustKernel(); double delta() const; };
class RobustKernelHuber : public RobustKernel {};
}  // namespace g2o
