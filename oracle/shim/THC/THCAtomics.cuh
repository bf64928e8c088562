// TEST INFRASTRUCTURE ONLY: empty shim, see ../ATen/ATen.h
#pragma once
