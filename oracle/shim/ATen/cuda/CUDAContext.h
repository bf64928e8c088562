// TEST INFRASTRUCTURE ONLY: empty shim, see ../ATen.h
#pragma once
