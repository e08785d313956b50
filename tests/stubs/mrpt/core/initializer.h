#pragma once  // stand-in (tests/stubs/README.md)
#define MRPT_INITIALIZER(f) static void f(); namespace { struct f##_reg { f##_reg() { f(); } } f##_reg_inst; } static void f()
